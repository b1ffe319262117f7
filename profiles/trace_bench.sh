#!/bin/bash
# kernel trace of one bench.py run -> per-kernel table (profiles/kstat.py).  usage: profiles/trace_bench.sh OUTDIR NKERNELS bench-args...
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; N=$2; shift 2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-mode --no-secondary "$@" > $OUT/bench.json 2> $OUT/bench.err
cd $GRAFT_REPO_ROOT && python3 profiles/kstat.py $OUT/trace $N | tee $OUT/kstat.txt
rm -rf $OUT/trace
