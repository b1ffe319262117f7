#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2e8; mkdir -p $OUT
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=5 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest.log
