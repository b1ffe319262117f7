#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2e2; mkdir -p $OUT
cd $R
timeout -k 10 300 python3 profiles/host_mode_rate.py > $OUT/hostmode_staged.json 2> $OUT/hostmode_staged.err; echo "hm rc=$?"
MDP_HOST_REGISTER=1 timeout -k 10 300 python3 profiles/host_mode_rate.py > $OUT/hostmode_registered.json 2> $OUT/hostmode_registered.err; echo "hm2 rc=$?"
cat $OUT/hostmode_staged.json $OUT/hostmode_registered.json
timeout -k 10 600 python -m pytest tests/test_gpu_rebomos.py tests/test_plugin_boundary.py tests/test_error_paths.py -m gpu -x -q 2>&1 | tail -3
