#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; O=gpurun_out/pt; mkdir -p $O
MDP_AEAM_PT_NSUB=3 MDP_PT_DBG=4 timeout -k 10 200 python3 bench.py --workload aeam --temp 863 --steps 2 --warmup 1 --no-cpu-baseline --no-host-mode > $O/dbg4.json 2> $O/dbg4.err
grep -h ptile $O/dbg4.err | tail -8
run() { tag=$1; shift
  env "$@" timeout -k 10 200 python3 bench.py --workload aeam --temp 863 --steps 100 --warmup 20 --no-cpu-baseline --no-host-mode > $O/bench_$tag.json 2> $O/bench_$tag.err
  python3 -c "
import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms'])"
}
run dbg0 MDP_AEAM_PT_NSUB=3
run dbg1 MDP_AEAM_PT_NSUB=3 MDP_PT_DBG=1
run dbg2 MDP_AEAM_PT_NSUB=3 MDP_PT_DBG=2
