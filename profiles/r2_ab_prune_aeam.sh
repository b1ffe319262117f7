#!/bin/bash
# A/B: dynamic pruning of the tile rows, aeam (MDP_PRUNE=0 switches it off)
set -u
cd $GRAFT_REPO_ROOT; O=gpurun_out/ab; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_aeam.py tests/test_gpu_domain.py tests/test_gpu_multirank.py tests/test_gpu_fullsize.py tests/test_gpu_resident.py -x -q -m gpu > $O/tests_prune_aeam.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests_prune_aeam.log
run() { tag=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py $ARGS --no-cpu-baseline --no-host-mode > $O/bench_$tag.json 2> $O/bench_$tag.err
  python3 profiles/print_bench.py $tag $O/bench_$tag.json
}
ARGS="--workload aeam --temp 863 --steps 1000 --warmup 20"
run aeam_adaptive X=1
run aeam_plain MDP_PRUNE=0
run aeam_b03 MDP_PRUNE_BUFFER=0.3
run aeam_b05 MDP_PRUNE_BUFFER=0.5
