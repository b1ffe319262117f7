#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=3 > gpurun_out/final/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/final/pytest.log
bash profiles/r2_collect.sh > gpurun_out/final/collect.log 2>&1; tail -4 gpurun_out/final/collect.log
