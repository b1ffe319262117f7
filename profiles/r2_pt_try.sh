#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; O=gpurun_out/pt; mkdir -p $O
MDP_AEAM_ROWS=32 timeout -k 10 900 python3 -m pytest tests/test_gpu_aeam.py tests/test_gpu_domain.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/tests32.log 2>&1; echo "tests rows=32 rc=$?"; tail -3 $O/tests32.log
MDP_AEAM_ROWS=32 MDP_AEAM_PERSIST=0 timeout -k 10 900 python3 -m pytest tests/test_gpu_aeam.py -x -q -m gpu > $O/tests32b.log 2>&1; echo "tests rows=32 gather kernels rc=$?"; tail -2 $O/tests32b.log
run() { tag=$1; shift
  env "$@" MDP_DEBUG=1 timeout -k 10 300 python3 bench.py --workload aeam --temp 863 --steps 300 --warmup 20 --no-cpu-baseline --no-host-mode > $O/bench_$tag.json 2> $O/bench_$tag.err
  grep "tile lists\|persistent" $O/bench_$tag.err | tail -2
  python3 -c "
import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms'], d['config'].get('pe_per_atom_end_eV'), d['config'].get('temp_end_K'), d['config'].get('reneighbor_wall_ms'))"
}
run rows16 X=1
run rows32 MDP_AEAM_ROWS=32
run rows32_ns4 MDP_AEAM_ROWS=32 MDP_AEAM_PT_NSUB=4
run rows32_gather MDP_AEAM_ROWS=32 MDP_AEAM_PERSIST=0
