#!/bin/bash
# quick per-kernel medians for the default bench (scratch helper)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/quick; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" > $OUT/trace.json 2> $OUT/trace.err || echo "trace failed"
cd $R
python3 profiles/summarize_trace.py $OUT/trace/*/*kernel_trace.csv | head -40
