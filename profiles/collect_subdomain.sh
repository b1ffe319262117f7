#!/bin/bash
# timeline and PMC traffic of rank 0's sub-domain for the 2-, 4- and 8-brick decompositions of the headline system
# (all bricks on one GPU as threads, rank 0 stepping alone at the end).  usage: bash profiles/collect_subdomain.sh r03
set -u
R=${1:-r04}; ROOT=$GRAFT_REPO_ROOT; STEPS=30
cd $ROOT; SHA=$(python3 -c "import bench; print(bench.kernel_source_sha())")
for N in 8 4 2; do
  OUT=$ROOT/gpurun_out/${R}_subdomain$N; rm -rf $OUT; mkdir -p $OUT
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/profiles/subdomain_step.py 24 $STEPS $N > $OUT/subdomain.json 2> $OUT/subdomain.err || echo "trace $N failed"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/profiles/subdomain_step.py 24 $STEPS $N > /dev/null 2> $OUT/pmc_fetch.err || echo "fetch $N failed"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/profiles/subdomain_step.py 24 $STEPS $N > /dev/null 2> $OUT/pmc_write.err || echo "write $N failed"
  cd $ROOT
  python3 profiles/step_timeline.py $OUT 20 > $OUT/timeline.txt 2>&1
  python3 profiles/pmc_subdomain_entry.py $OUT rebomos:24x24x24:$N $SHA $STEPS > $OUT/pmc_entry.json 2> $OUT/pmc_entry.err
  echo "== $N bricks"; cat $OUT/subdomain.json; grep "mean of the last" $OUT/timeline.txt; python3 -c "
import json;d=json.load(open('$OUT/pmc_entry.json'));[print(k, v['bytes_per_step']) for k,v in d.items()]"
done
