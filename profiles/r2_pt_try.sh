#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT; O=gpurun_out/pt; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
MDP_AEAM_PERSIST_FORCE=1 timeout -k 10 400 python3 -m pytest tests/test_gpu_aeam.py -x -q -m gpu > $O/tests_pf.log 2>&1; echo "tests (persistent force) rc=$?"; tail -2 $O/tests_pf.log
run() { tag=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --workload aeam --temp 863 --steps 1000 --warmup 20 --no-cpu-baseline --no-host-mode > $O/bench_$tag.json 2> $O/bench_$tag.err
  python3 -c "
import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms'], d['config'].get('pe_per_atom_end_eV'), d['config'].get('temp_end_K'))"
}
run new X=1
run old MDP_AEAM_PERSIST=0
