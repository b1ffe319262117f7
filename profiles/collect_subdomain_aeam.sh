#!/bin/bash
# aeam: timeline and PMC traffic of rank 0's sub-domain for the 2-, 4- and 8-brick decompositions of the 16.1 M-atom
# alloy (BASELINE.json configs[4]: fcc 159^3 cells, 0.75 % Si, 863 K), all bricks on one GPU as threads, rank 0
# stepping alone at the end (profiles/subdomain_step.py).  usage: bash profiles/collect_subdomain_aeam.sh r04 [nrep] [bricks...]
set -u
R=${1:-r04}; NREP=${2:-159}; shift; shift; BR=${@:-8 4 2}
ROOT=$GRAFT_REPO_ROOT; STEPS=30
cd $ROOT; SHA=$(python3 -c "import bench; print(bench.kernel_source_sha())")
for N in $BR; do
  OUT=$ROOT/gpurun_out/${R}_aeam_subdomain$N; rm -rf $OUT; mkdir -p $OUT
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/profiles/subdomain_step.py $NREP $STEPS $N aeam > $OUT/subdomain.json 2> $OUT/subdomain.err || echo "trace $N failed"
  echo "== aeam $N bricks"; cat $OUT/subdomain.json
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/profiles/subdomain_step.py $NREP $STEPS $N aeam > /dev/null 2> $OUT/pmc_fetch.err || echo "fetch $N failed"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/profiles/subdomain_step.py $NREP $STEPS $N aeam > /dev/null 2> $OUT/pmc_write.err || echo "write $N failed"
  cd $ROOT
  python3 profiles/step_timeline.py $OUT 20 > $OUT/timeline.txt 2>&1
  python3 profiles/pmc_subdomain_entry.py $OUT aeam:${NREP}x${NREP}x${NREP}:$N $SHA $STEPS > $OUT/pmc_entry.json 2> $OUT/pmc_entry.err
  grep "mean of the last" $OUT/timeline.txt; python3 -c "
import json;d=json.load(open('$OUT/pmc_entry.json'));[print(k, v['bytes_per_step']) for k,v in d.items()]"
  rm -rf $OUT/pmc_fetch/*/*.db $OUT/pmc_write/*/*.db $OUT/trace/*/*.db 2>/dev/null
done
