#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2e2; mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
timeout -k 10 300 python3 profiles/host_mode_rate.py > $OUT/hostmode_staged.json 2> $OUT/hostmode_staged.err; echo "hm rc=$?"
MDP_HOST_REGISTER=1 timeout -k 10 300 python3 profiles/host_mode_rate.py > $OUT/hostmode_registered.json 2> $OUT/hostmode_registered.err; echo "hm2 rc=$?"
cat $OUT/hostmode_staged.json $OUT/hostmode_registered.json
nproc; lscpu | grep -E "Model name|^CPU\(s\)"
