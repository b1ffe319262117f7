#!/bin/bash
# multi-rank rehearsal of bench.py on ONE GPU (gloo-staged halo; <= 4 ranks share the card)
set -u
cd $GRAFT_REPO_ROOT; O=gpurun_out/rehearse; mkdir -p $O
run() { tag=$1; n=$2; shift 2
  MDP_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 280 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $n "$@" --no-cpu-baseline --no-host-mode > $O/$tag.out 2> $O/$tag.err
  echo "$tag rc=$?"; grep '^{' $O/$tag.out | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('  ', d['value'], d['ms_per_step'], d['n_gpus'], d['config']['parallelism'], d['config'].get('pe_per_atom_end_eV'), d['config'].get('temp_end_K'), 'reneigh', d['config'].get('reneighborings_in_timed_region'))"
}
run rebo1 1 --replicate 10 10 10 --temp 300 --steps 40 --warmup 5
run rebo2 2 --replicate 10 10 10 --temp 300 --steps 40 --warmup 5
run rebo4 4 --replicate 10 10 10 --temp 300 --steps 40 --warmup 5
run aeam1 1 --workload aeam --replicate 30 30 30 --temp 863 --steps 60 --warmup 5
run aeam4 4 --workload aeam --replicate 30 30 30 --temp 863 --steps 60 --warmup 5
