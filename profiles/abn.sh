#!/bin/bash
# the bench line under several environments, twice each, alternating.  usage: profiles/abn.sh "<VAR=a>" "<VAR=b>" ... [-- bench args]
set -u
ENVS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done; [ $# -gt 0 ] && shift
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
for rep in 1 2; do for v in "${ENVS[@]}"; do
  env $v timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-host-mode "$@" > gpurun_out/ab/out.json 2> gpurun_out/ab/out.err
  python3 -c "
import json;d=json.load(open('gpurun_out/ab/out.json'));print('$v', d['value'], d['ms_per_step'], {k[:22]: v for k, v in d['roofline']['phase_ms'].items()})"
done; done
