#!/bin/bash
# Collect the judged artifacts for one workload on the GPU box:
#   bench JSON, rocprofv3 --kernel-trace --stats of the same command, FETCH_SIZE / WRITE_SIZE PMC passes (own runs).
# usage: bash profiles/collect.sh <tag> <pmc key, e.g. rebomos:24x24x24:1> [bench args...]   (outputs under gpurun_out/<tag>/)
set -u
TAG=$1; KEY=$2; shift 2
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
cd $R
python3 bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err || echo "bench failed"
SHA=$(python3 -c "import bench; print(bench.kernel_source_sha())")
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py "$@" --no-cpu-baseline --no-host-mode --no-secondary > $OUT/trace.json 2> $OUT/trace.err || echo "trace failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline --no-host-mode --no-secondary > /dev/null 2> $OUT/pmc_fetch.err || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline --no-host-mode --no-secondary > /dev/null 2> $OUT/pmc_write.err || echo "write failed"
cd $R
python3 profiles/summarize_trace.py $OUT/trace/*/*kernel_trace.csv > $OUT/kernel_summary.txt
cp $OUT/trace/*/*kernel_stats.csv $OUT/rocprofv3_kernel_stats.csv 2>/dev/null
python3 profiles/pmc_traffic_entry.py $OUT "$KEY" "$SHA" > $OUT/pmc_entry.json
cat $OUT/bench.json; head -14 $OUT/kernel_summary.txt; head -40 $OUT/pmc_entry.json
