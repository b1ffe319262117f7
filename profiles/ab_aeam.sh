#!/bin/bash
# AEAM configuration #3 (1 000 188 atoms, 863 K) under several environments, one bench line each.
# usage: profiles/ab_aeam.sh OUTDIR STEPS "VAR=a" "VAR=b VAR2=c" ...   (settings are scoped to their one run)
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; STEPS=$2; shift 2
mkdir -p $OUT
i=0
for v in "$@"; do
  i=$((i+1))
  ( [ "$v" != "-" ] && export $v
    timeout -k 10 400 python3 bench.py --workload aeam --temp 863 --steps $STEPS --warmup 20 --no-cpu-baseline --no-host-mode --no-secondary \
       > $OUT/run$i.json 2> $OUT/run$i.err ) || echo "run $i failed"
  echo "== $v"
  python3 - $OUT/run$i.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print("value",d["value"],"ms/step",d["ms_per_step"],"path_ms",r.get("path_ms"),"phase_ms",r.get("phase_ms"),"reneigh",d["config"].get("reneighborings_in_timed_region"),"prunes",d["config"].get("row_prunings_in_timed_region_rank0"),"pe_end",d["config"].get("pe_per_atom_end_eV"),"T",d["config"].get("temp_end_K"))
except Exception as e:
    print("no line:",e)
PY
  grep -h "persistent" $OUT/run$i.err | head -3
done
