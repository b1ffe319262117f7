#!/bin/bash
# The judged artifacts of a round: bench JSON (default run with the secondary block) + rocprofv3 kernel stats + PMC
# traffic per configuration, sub-domain timeline.  usage (on the GPU box): bash profiles/collect_round.sh r03
# Outputs under gpurun_out/<round>_*; `python3 profiles/store_round.py <round>` (here) copies the summaries into
# profiles/ and refreshes profiles/pmc_traffic.json.
set -u
R=${1:-r03}
cd $GRAFT_REPO_ROOT
bash profiles/collect.sh ${R}_rebomos4m rebomos:24x24x24:1 --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_rebomos4m.log 2>&1; tail -2 gpurun_out/${R}_rebomos4m.log | cut -c1-300
bash profiles/collect.sh ${R}_aeam1m aeam:63x63x63:1 --workload aeam --temp 863 --steps 1000 --warmup 20 > gpurun_out/${R}_aeam1m.log 2>&1; tail -2 gpurun_out/${R}_aeam1m.log | cut -c1-300
bash profiles/collect.sh ${R}_rebomos4m_300K rebomos:24x24x24:1 --temp 300 --steps 600 --warmup 20 > gpurun_out/${R}_rebomos4m_300K.log 2>&1; tail -2 gpurun_out/${R}_rebomos4m_300K.log | cut -c1-300
OUT=$GRAFT_REPO_ROOT/gpurun_out/${R}_subdomain8; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/profiles/subdomain_step.py 24 40 > $OUT/subdomain.json 2> $OUT/subdomain.err
cd $GRAFT_REPO_ROOT
python3 profiles/step_timeline.py $OUT > $OUT/timeline.txt 2>&1; cat $OUT/subdomain.json; tail -3 $OUT/timeline.txt
