#!/bin/bash
# A/B helper: aeam parity tests + the 1 M atom bench line
set -u
cd $GRAFT_REPO_ROOT; O=gpurun_out/ab; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_aeam.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
timeout -k 10 300 python3 bench.py --workload aeam --temp 863 --steps 1000 --warmup 20 --no-cpu-baseline --no-host-mode > $O/bench.json 2> $O/bench.err
python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms'], 'reneighbor_ms', d['config'].get('reneighbor_wall_ms'))"
