#!/bin/bash
# timelines of the 12x12x12 / 300 K REBO-MoS sub-domain: plain one-GPU run and one-rank RCCL rehearsal (see trace_self.sh)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MASTER_ADDR=127.0.0.1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
one() { tag=$1; shift
  mkdir -p $OUT/$tag
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$tag/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-mode --no-secondary "$@" > $OUT/$tag/bench.json 2> $OUT/$tag/bench.err
  echo "== $tag rc=$?"; python3 $GRAFT_REPO_ROOT/profiles/step_timeline.py $OUT/$tag 60 45 30 > $OUT/$tag/timeline.txt 2>&1; head -${LINES_SHOWN:-22} $OUT/$tag/timeline.txt
  rm -rf $OUT/$tag/trace
}
R="--replicate 12 12 12 --temp 300 --steps 120 --warmup 10"
MDP_BENCH_SELF_REMOTE=1 MASTER_PORT=29561 one self_rebo $R
one plain_rebo $R
