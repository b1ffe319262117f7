#!/bin/bash
# round-2 judged artifacts: bench JSON + rocprofv3 kernel stats + PMC traffic per configuration, sub-domain timeline
set -u
cd $GRAFT_REPO_ROOT
bash profiles/collect.sh r02_rebomos4m rebomos:24x24x24:1 --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_rebomos4m.log 2>&1; tail -3 gpurun_out/r02_rebomos4m.log
bash profiles/collect.sh r02_aeam1m aeam:63x63x63:1 --workload aeam --temp 863 --steps 1000 --warmup 20 > gpurun_out/r02_aeam1m.log 2>&1; tail -3 gpurun_out/r02_aeam1m.log
bash profiles/collect.sh r02_aeam16m aeam:159x159x159:1 --workload aeam --replicate 159 159 159 --temp 863 --steps 100 --warmup 10 > gpurun_out/r02_aeam16m.log 2>&1; tail -3 gpurun_out/r02_aeam16m.log
bash profiles/collect.sh r02_rebomos4m_300K rebomos:24x24x24:1 --temp 300 --steps 600 --warmup 20 > gpurun_out/r02_rebomos4m_300K.log 2>&1; tail -3 gpurun_out/r02_rebomos4m_300K.log
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02_subdomain8; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/profiles/subdomain_step.py 24 40 > $OUT/subdomain.json 2> $OUT/subdomain.err
cd $GRAFT_REPO_ROOT
python3 profiles/step_timeline.py $OUT > $OUT/timeline.txt 2>&1; cat $OUT/subdomain.json; tail -3 $OUT/timeline.txt
