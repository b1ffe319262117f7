#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; O=gpurun_out/pt; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_aeam.py tests/test_gpu_domain.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
run() { tag=$1; shift
  timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline --no-host-mode > $O/bench_$tag.json 2> $O/bench_$tag.err
  python3 -c "
import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms'], 'reneighbor_ms', d['config'].get('reneighbor_wall_ms'))"
}
run aeam --workload aeam --temp 863 --steps 1000 --warmup 20
