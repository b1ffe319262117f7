#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2e1; mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
timeout -k 10 400 python3 bench.py --workload aeam --replicate 159 159 159 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/aeam16m.json 2> $OUT/aeam16m.err; echo "aeam16m rc=$?"
timeout -k 10 300 python3 bench.py --workload aeam --temp 863 --steps 1000 --warmup 20 --check-every 1 --no-cpu-baseline > $OUT/aeam1m_863.json 2> $OUT/aeam1m_863.err; echo "aeam1m rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace05 -- python3 $R/bench.py --replicate 12 12 12 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/rebo05.json 2> $OUT/rebo05.err; echo "rebo05 rc=$?"
cd $R
mkdir -p $OUT/t05 && cp -r $OUT/trace05 $OUT/t05/trace
python3 profiles/step_timeline.py $OUT/t05 > $OUT/rebo05_timeline.txt 2>&1
cat $OUT/aeam16m.json $OUT/aeam1m_863.json $OUT/rebo05.json; tail -3 $OUT/rebo05_timeline.txt
