#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2e9; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_domain.py tests/test_gpu_multirank.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
MDP_HALO_OVERLAP=lj timeout -k 10 600 python -m pytest tests/test_gpu_domain.py -x -q -k "bricks" > $OUT/pytest_lj.log 2>&1; echo "pytest(lj overlap) rc=$?"; tail -3 $OUT/pytest_lj.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/profiles/subdomain_step.py 24 40 > $OUT/subdomain.json 2> $OUT/subdomain.err; echo "rc=$?"
cd $R
cat $OUT/subdomain.json
python3 profiles/step_timeline.py $OUT > $OUT/timeline.txt 2>&1; cat $OUT/timeline.txt
