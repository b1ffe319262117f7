#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2e7; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload aeam --temp 863 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/trace.json 2> $OUT/trace.err || echo "trace failed"
cd $R
python3 profiles/summarize_trace.py $OUT/trace/*/*kernel_trace.csv | head -45
