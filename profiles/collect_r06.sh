#!/bin/bash
# round-6 collection in pieces that fit one gpurun call each.  usage: bash profiles/collect_r06.sh <piece>
#   order: bench (kernel stats + PMC traffic) -> python3 profiles/store_round.py r05 HERE -> lines (the default bench
#   command and the two 300 K / 863 K configurations again: their stored JSON then carries `traffic`)
set -u
cd $GRAFT_REPO_ROOT; R=r06
slim() { for t in "$@"; do rm -rf gpurun_out/$t/trace gpurun_out/$t/pmc_fetch gpurun_out/$t/pmc_write; done; }  # (gpurun merges at most 64 MiB back)
case "${1:-}" in
 bench)   # kernel stats and PMC traffic per configuration
   bash profiles/collect.sh ${R}_rebomos4m rebomos:24x24x24:1 --gpus 1 --steps 20 --warmup 5 --no-secondary > gpurun_out/${R}_rebomos4m.log 2>&1; tail -2 gpurun_out/${R}_rebomos4m.log | cut -c1-300
   bash profiles/collect.sh ${R}_aeam1m aeam:63x63x63:1 --workload aeam --temp 863 --steps 1000 --warmup 20 > gpurun_out/${R}_aeam1m.log 2>&1; tail -2 gpurun_out/${R}_aeam1m.log | cut -c1-300
   bash profiles/collect.sh ${R}_rebomos4m_300K rebomos:24x24x24:1 --temp 300 --steps 600 --warmup 20 > gpurun_out/${R}_rebomos4m_300K.log 2>&1; tail -2 gpurun_out/${R}_rebomos4m_300K.log | cut -c1-300; slim ${R}_rebomos4m ${R}_aeam1m ${R}_rebomos4m_300K ;;
 offlattice) # the two off-lattice secondary configurations by themselves: kernel stats of the same commands
   bash profiles/collect.sh ${R}_rebomos4m_strained rebomos_strained:24x24x24:1 --strain 1.12 0.15 --temp 300 --steps 200 --warmup 20 > gpurun_out/${R}_rebomos4m_strained.log 2>&1; tail -2 gpurun_out/${R}_rebomos4m_strained.log | cut -c1-300
   bash profiles/collect.sh ${R}_aeam1m_8pct aeam_8pct:63x63x63:1 --workload aeam --frac2 0.08 --temp 863 --steps 300 --warmup 20 > gpurun_out/${R}_aeam1m_8pct.log 2>&1; tail -2 gpurun_out/${R}_aeam1m_8pct.log | cut -c1-300; slim ${R}_rebomos4m_strained ${R}_aeam1m_8pct ;;
 lines)   # AFTER store_round.py: the driver's default command and the other configurations, lines with `traffic`
   mkdir -p gpurun_out/${R}_lines
   python3 bench.py > gpurun_out/${R}_lines/default.json 2> gpurun_out/${R}_lines/default.err; tail -c 400 gpurun_out/${R}_lines/default.json
   python3 bench.py --workload aeam --temp 863 --steps 1000 --warmup 20 --no-secondary > gpurun_out/${R}_lines/aeam1m.json 2> gpurun_out/${R}_lines/aeam1m.err
   python3 bench.py --temp 300 --steps 600 --warmup 20 --no-secondary > gpurun_out/${R}_lines/rebomos4m_300K.json 2> gpurun_out/${R}_lines/rebomos4m_300K.err
   for t in default aeam1m rebomos4m_300K; do python3 profiles/print_bench.py $t gpurun_out/${R}_lines/$t.json 2>/dev/null | cut -c1-400; done ;;
 aeam16m) bash profiles/collect.sh ${R}_aeam16m aeam:159x159x159:1 --workload aeam --replicate 159 159 159 --temp 863 --steps 100 --warmup 10 > gpurun_out/${R}_aeam16m.log 2>&1; tail -2 gpurun_out/${R}_aeam16m.log | cut -c1-300; slim ${R}_aeam16m ;;
 sub)     bash profiles/collect_subdomain.sh $R > gpurun_out/${R}_collect_sub.log 2>&1; tail -12 gpurun_out/${R}_collect_sub.log ;;
 subaeam) bash profiles/collect_subdomain_aeam.sh $R 159 ${2:-8 4 2} > gpurun_out/${R}_collect_subaeam.log 2>&1; tail -12 gpurun_out/${R}_collect_subaeam.log ;;
 rehearse) bash profiles/rehearse.sh ${R}_rehearse_r5style > gpurun_out/${R}_rehearse_r5style.log 2>&1; tail -30 gpurun_out/${R}_rehearse_r5style.log; bash profiles/rehearse_r06.sh ${R}_rehearse > gpurun_out/${R}_rehearse.log 2>&1; cat gpurun_out/${R}_rehearse.log ;;
 counters) bash profiles/pmc_passes.sh ${R}_pmc_rebomos --no-secondary --no-host-mode > gpurun_out/${R}_pmc_rebomos.log 2>&1
           python3 profiles/summarize_pmc.py gpurun_out/${R}_pmc_rebomos > gpurun_out/${R}_rebomos4m_pmc_sq_tcp_counters.json
           bash profiles/pmc_passes.sh ${R}_pmc_aeam --workload aeam --temp 863 --no-secondary --no-host-mode > gpurun_out/${R}_pmc_aeam.log 2>&1
           python3 profiles/summarize_pmc.py gpurun_out/${R}_pmc_aeam > gpurun_out/${R}_aeam1m_pmc_sq_tcp_counters.json
           bash profiles/pmc_passes.sh ${R}_pmc_strained --strain 1.12 0.15 --temp 300 --warmup 150 --no-secondary --no-host-mode > gpurun_out/${R}_pmc_strained.log 2>&1
           python3 profiles/summarize_pmc.py gpurun_out/${R}_pmc_strained > gpurun_out/${R}_rebomos4m_strained_pmc_sq_tcp_counters.json
           rm -rf gpurun_out/${R}_pmc_rebomos gpurun_out/${R}_pmc_aeam gpurun_out/${R}_pmc_strained; head -c 600 gpurun_out/${R}_rebomos4m_strained_pmc_sq_tcp_counters.json ;;
 pin)     timeout -k 10 1000 python3 profiles/trajectory_pin.py 1000 > gpurun_out/${R}_trajectory_pin.json 2> gpurun_out/${R}_trajectory_pin.err; python3 -c "
import json;d=json.load(open('gpurun_out/${R}_trajectory_pin.json'))
for k,v in d.items(): print(k, {a:b for a,b in v.items() if not a.startswith('etotal')}, 'E drift dev', v['etotal_per_atom_device'][-1]-v['etotal_per_atom_device'][0], 'host', v['etotal_per_atom_host'][-1]-v['etotal_per_atom_host'][0])" ;;
 *) echo "pieces: bench offlattice lines aeam16m sub subaeam [bricks] rehearse counters pin" ;;
esac
