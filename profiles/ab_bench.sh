#!/bin/bash
# same-box A/B of two builds of the library on the headline line: profiles/ab_bench.sh BASE.so [bench args]
# (A = MDP_LIB_PATH=BASE.so, B = the tree's library; two rounds each, alternating)
set -u
BASE=$1; shift
for r in 1 2; do
  for v in A B; do
    if [ $v = A ]; then export MDP_LIB_PATH=$BASE; else unset MDP_LIB_PATH; fi
    python3 bench.py --no-secondary --no-cpu-baseline --no-host-mode "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v$r', d['ms_per_step'], d['value'], {k[:28]:v for k,v in d['roofline']['phase_ms'].items()}, d['config'].get('reneighbor_wall_ms'))"
  done
done
