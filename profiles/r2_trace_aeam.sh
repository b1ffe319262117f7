#!/bin/bash
# kernel trace of the aeam 1 M atom bench (where does a reneighboring go?)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/aeam_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload aeam --temp 863 --steps 1000 --warmup 20 --no-cpu-baseline --no-host-mode > $OUT/trace.json 2> $OUT/trace.err || echo "trace failed"
cd $R
python3 profiles/summarize_trace.py $OUT/trace/*/*kernel_trace.csv > $OUT/kernel_summary.txt
head -50 $OUT/kernel_summary.txt
