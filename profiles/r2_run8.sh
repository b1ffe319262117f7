#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2e8; mkdir -p $OUT
cd $R
timeout -k 10 1100 python -m pytest tests/test_gpu_aeam.py tests/test_plugin_boundary.py tests/test_error_paths.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest.log
