#!/bin/bash
# A/B: dynamic pruning of the Lennard-Jones rows (MDP_PRUNE=0 switches it off)
set -u
cd $GRAFT_REPO_ROOT; O=gpurun_out/ab; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_rebomos.py tests/test_gpu_edge.py tests/test_gpu_resident.py tests/test_gpu_domain.py tests/test_gpu_multirank.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/tests_prune.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests_prune.log
run() { tag=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py $ARGS --no-cpu-baseline --no-host-mode > $O/bench_$tag.json 2> $O/bench_$tag.err
  python3 profiles/print_bench.py $tag $O/bench_$tag.json
}
ARGS="--steps 40 --warmup 5"
run cold_adaptive X=1
run cold_b02 MDP_PRUNE_BUFFER=0.2
run cold_plain MDP_PRUNE=0
ARGS="--temp 300 --steps 600 --warmup 20"
run 300K_adaptive X=1
run 300K_b05 MDP_PRUNE_BUFFER=0.5
run 300K_b06 MDP_PRUNE_BUFFER=0.6
run 300K_plain MDP_PRUNE=0
