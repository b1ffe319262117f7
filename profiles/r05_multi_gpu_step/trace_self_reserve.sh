#!/bin/bash
# (MDP_COMM_RESERVE_CUS was an experiment of round 5 -- a compute stream with a CU mask -- and is no longer in the library;
#  3_reserved_cus_8_16.log is what it measured)
# as trace_self.sh, self-remote runs only, with MDP_COMM_RESERVE_CUS=$2 (compute units the compute stream leaves free)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MASTER_ADDR=127.0.0.1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MDP_BENCH_SELF_REMOTE=1
one() { tag=$1; shift
  mkdir -p $OUT/$tag
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$tag/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-mode --no-secondary "$@" > $OUT/$tag/bench.json 2> $OUT/$tag/bench.err
  echo "== $tag rc=$?"; python3 $GRAFT_REPO_ROOT/profiles/step_timeline.py $OUT/$tag 60 45 30 > $OUT/$tag/timeline.txt 2>&1; head -24 $OUT/$tag/timeline.txt
  rm -rf $OUT/$tag/trace
}
R="--replicate 12 12 12 --temp 300 --steps 120 --warmup 10"
A="--workload aeam --replicate 63 63 63 --temp 863 --steps 120 --warmup 10"
for n in $2; do
  [ "$n" = 0 ] && unset MDP_COMM_RESERVE_CUS
  MDP_COMM_RESERVE_CUS=$n MASTER_PORT=29561 one self_rebo_res$n $R
  MDP_COMM_RESERVE_CUS=$n MASTER_PORT=29562 one self_aeam_res$n $A
done
