#!/bin/bash
# round-4 collection in pieces that fit one gpurun call each.  usage: bash profiles/collect_r04.sh <piece>
set -u
cd $GRAFT_REPO_ROOT; R=r04
case "${1:-}" in
 bench)   # the driver's command (default run with the secondary block), kernel stats and PMC traffic per configuration
   bash profiles/collect.sh ${R}_rebomos4m rebomos:24x24x24:1 --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_rebomos4m.log 2>&1; tail -2 gpurun_out/${R}_rebomos4m.log | cut -c1-300
   bash profiles/collect.sh ${R}_aeam1m aeam:63x63x63:1 --workload aeam --temp 863 --steps 1000 --warmup 20 > gpurun_out/${R}_aeam1m.log 2>&1; tail -2 gpurun_out/${R}_aeam1m.log | cut -c1-300
   bash profiles/collect.sh ${R}_rebomos4m_300K rebomos:24x24x24:1 --temp 300 --steps 600 --warmup 20 > gpurun_out/${R}_rebomos4m_300K.log 2>&1; tail -2 gpurun_out/${R}_rebomos4m_300K.log | cut -c1-300 ;;
 aeam16m) bash profiles/collect.sh ${R}_aeam16m aeam:159x159x159:1 --workload aeam --replicate 159 159 159 --temp 863 --steps 100 --warmup 10 > gpurun_out/${R}_aeam16m.log 2>&1; tail -2 gpurun_out/${R}_aeam16m.log | cut -c1-300 ;;
 sub)     bash profiles/collect_subdomain.sh $R > gpurun_out/${R}_collect_sub.log 2>&1; tail -12 gpurun_out/${R}_collect_sub.log ;;
 subaeam) bash profiles/collect_subdomain_aeam.sh $R 159 ${2:-8 4 2} > gpurun_out/${R}_collect_subaeam.log 2>&1; tail -12 gpurun_out/${R}_collect_subaeam.log ;;
 rehearse) bash profiles/rehearse.sh ${R}_rehearse > gpurun_out/${R}_rehearse.log 2>&1; cat gpurun_out/${R}_rehearse.log ;;
 counters) bash profiles/pmc_passes.sh ${R}_pmc_rebomos --no-secondary --no-host-mode > gpurun_out/${R}_pmc_rebomos.log 2>&1
           python3 profiles/summarize_pmc.py gpurun_out/${R}_pmc_rebomos > gpurun_out/${R}_rebomos4m_pmc_sq_tcp_counters.json
           bash profiles/pmc_passes.sh ${R}_pmc_aeam --workload aeam --temp 863 --no-secondary --no-host-mode > gpurun_out/${R}_pmc_aeam.log 2>&1
           python3 profiles/summarize_pmc.py gpurun_out/${R}_pmc_aeam > gpurun_out/${R}_aeam1m_pmc_sq_tcp_counters.json
           rm -rf gpurun_out/${R}_pmc_*/*/*/*.db; head -c 600 gpurun_out/${R}_rebomos4m_pmc_sq_tcp_counters.json ;;
 pin)     timeout -k 10 1000 python3 profiles/trajectory_pin.py 1000 > gpurun_out/${R}_trajectory_pin.json 2> gpurun_out/${R}_trajectory_pin.err; python3 -c "
import json;d=json.load(open('gpurun_out/${R}_trajectory_pin.json'))
for k,v in d.items(): print(k, {a:b for a,b in v.items() if not a.startswith('etotal')}, 'E drift dev', v['etotal_per_atom_device'][-1]-v['etotal_per_atom_device'][0], 'host', v['etotal_per_atom_host'][-1]-v['etotal_per_atom_host'][0])" ;;
 *) echo "pieces: bench aeam16m sub subaeam [bricks] rehearse counters pin" ;;
esac
