#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2e4; mkdir -p $OUT
cd $R
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/rebo4m.json 2> $OUT/rebo4m.err; echo "rebo4m rc=$?"
timeout -k 10 300 python3 bench.py --workload aeam --temp 863 --steps 1000 --warmup 20 --no-cpu-baseline > $OUT/aeam1m_863.json 2> $OUT/aeam1m_863.err; echo "aeam1m rc=$?"
timeout -k 10 300 python3 bench.py --temp 300 --steps 600 --warmup 20 --no-cpu-baseline --no-host-mode > $OUT/rebo4m_300.json 2> $OUT/rebo4m_300.err; echo "rebo300 rc=$?"
timeout -k 10 400 python3 bench.py --workload aeam --replicate 159 159 159 --temp 863 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/aeam16m.json 2> $OUT/aeam16m.err; echo "aeam16m rc=$?"
cat $OUT/*.json; tail -3 $OUT/*.err
