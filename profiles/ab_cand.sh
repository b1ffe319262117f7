#!/bin/bash
# A/B of the candidate look counts (MDP_CAND_PRUNE=0: rows looked at in full), kernel stats of the headline run
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/ab_cand; mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in 1 0; do
  export MDP_CAND_PRUNE=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/v$v -- python3 bench.py --steps 40 --warmup 10 --no-secondary --no-cpu-baseline --no-host-mode > $O/v$v.json 2> $O/v$v.err
  f=$(find $O/v$v -name "*kernel_stats.csv" | head -1)
  echo "== MDP_CAND_PRUNE=$v"; python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:7]:
    print("  %-60s calls %5s avg_us %9.1f total_ms %8.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
  rm -rf $O/v$v
done
