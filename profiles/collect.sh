#!/bin/bash
# Collect the judged artifacts for one workload on the GPU box:
#   bench JSON, rocprofv3 --kernel-trace --stats of the same command, FETCH_SIZE / WRITE_SIZE PMC passes.
# usage: bash profiles/collect.sh <tag> [bench args...]      (outputs under gpurun_out/<tag>/)
set -u
TAG=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
python3 $R/bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err || echo "bench failed"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py "$@" --no-cpu-baseline > $OUT/trace.json 2> $OUT/trace.err || echo "trace failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py "$@" --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py "$@" --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err || echo "write failed"
cd $R
python3 profiles/summarize_trace.py $OUT/trace/*/*kernel_trace.csv > $OUT/kernel_summary.txt
python3 profiles/summarize_pmc.py $OUT > $OUT/pmc_summary.json
cat $OUT/bench.json; head -12 $OUT/kernel_summary.txt; cat $OUT/pmc_summary.json
