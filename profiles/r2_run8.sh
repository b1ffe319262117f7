#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2e8; mkdir -p $OUT
cd $R
timeout -k 10 1100 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_domain.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
export MDP_BENCH_BACKEND=gloo
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 40 --warmup 10 --replicate 8 8 8 --temp 300 --check-every 5 > $OUT/bench_n2_gloo.json 2> $OUT/bench_n2_gloo.err; echo "bench n2 rc=$?"
python3 -c "
import json; d=json.load(open('$OUT/bench_n2_gloo.json')); print(d['value'], d['ms_per_step'], d['config']['reneighbor_wall_ms'], d['config']['transport'])"
