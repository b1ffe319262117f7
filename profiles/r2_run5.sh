#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2e5; mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_aeam.py tests/test_gpu_domain.py tests/test_gpu_edge.py tests/test_gpu_fullsize.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
timeout -k 10 300 python3 bench.py --workload aeam --temp 863 --steps 1000 --warmup 20 --no-cpu-baseline > $OUT/aeam1m_863.json 2> $OUT/aeam1m_863.err; echo "aeam1m rc=$?"
python3 - <<PY
import json
d=json.load(open("$OUT/aeam1m_863.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms"], d["config"]["reneighborings_in_timed_region"], d["config"]["reneighbor_wall_ms"])
PY
