#!/bin/bash
# A/B of the bench line under two environments, alternating, on the GPU box.
# usage: profiles/ab.sh "<VAR=a ...>" "<VAR=b ...>" [bench args...]      e.g. profiles/ab.sh MDP_PRUNE=1 MDP_PRUNE=0
set -u
A=$1; B=$2; shift 2
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
for rep in 1 2; do for v in "$A" "$B"; do
  env $v timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-host-mode "$@" > gpurun_out/ab/out.json 2> gpurun_out/ab/out.err
  python3 -c "
import json;d=json.load(open('gpurun_out/ab/out.json'));print('$v', d['value'], d['ms_per_step'], d['roofline']['phase_ms'])"
done; done
