#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2e3; mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_domain.py -x -q -s -k rccl > $OUT/pytest_rccl.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|^Thread\|^$\|^Extension\|^Current thread" $OUT/pytest_rccl.log | tail -40
