#!/bin/bash
# A/B helper: rebomos parity tests + the 3.98 M atom bench line (extra environment as arguments: VAR=value ...)
set -u
cd $GRAFT_REPO_ROOT; O=gpurun_out/ab; mkdir -p $O
env "$@" timeout -k 10 900 python3 -m pytest tests/test_gpu_rebomos.py tests/test_gpu_edge.py -x -q -m gpu > $O/tests_rebo.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests_rebo.log
for rep in 1 2; do
env "$@" timeout -k 10 300 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-host-mode > $O/bench_rebo.json 2> $O/bench_rebo.err
python3 -c "
import json; d=json.load(open('$O/bench_rebo.json')); print('with', '$*', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms'])"
timeout -k 10 300 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-host-mode > $O/bench_rebo0.json 2> $O/bench_rebo0.err
python3 -c "
import json; d=json.load(open('$O/bench_rebo0.json')); print('plain', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms'])"
done
