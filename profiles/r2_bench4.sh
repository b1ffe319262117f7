#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; O=gpurun_out/final; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r02_rebomos4m_bench.json 2> $O/a.err
python3 bench.py --workload aeam --temp 863 --steps 1000 --warmup 20 > $O/r02_aeam1m_bench.json 2> $O/b.err
python3 bench.py --workload aeam --replicate 159 159 159 --temp 863 --steps 100 --warmup 10 > $O/r02_aeam16m_bench.json 2> $O/c.err
python3 bench.py --temp 300 --steps 600 --warmup 20 > $O/r02_rebomos4m_300K_bench.json 2> $O/d.err
for f in $O/r02_*_bench.json; do python3 -c "
import json,sys; d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d['roofline']['traffic'], d['roofline']['frac'])"; done
