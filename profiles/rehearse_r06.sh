#!/bin/bash
# Round 6 rehearsals on ONE GPU.  usage: profiles/rehearse_r06.sh <tag> ; output under gpurun_out/<tag>/
#  (1) `python3 bench.py --gpus 2|4` as the driver starts it -- own rank processes, the library's transport, two-call
#      steps, overlap-policy trial -- with the ranks sharing the card through the RCCL TEST DOUBLE (tests/native):
#      a rehearsal of the schedule with peers that are not the rank itself; its timings are no measurement.
#  (2) one rank through REAL RCCL to itself (MDP_BENCH_SELF_REMOTE=1), the plain one-GPU run of the same system beside
#      it: what the N>1 code path costs per step on the device, and which overlap policy the trial picks on this box.
set -u
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-rehearse_r06}; mkdir -p $O
DOUBLE=$GRAFT_REPO_ROOT/tests/native/libfake_rccl.so
show() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); c=d['config']
print('  ', d['value'], 'Matom-steps/s', d['ms_per_step'], 'ms  n_gpus', d['n_gpus'], 'rccl_ranks', c.get('rccl_ranks'), '|', c['transport'][:70], '| policy', c.get('overlap_policy'), '| PE/atom', c.get('pe_per_atom_end_eV'), 'T', c.get('temp_end_K'), '| reneigh', c.get('reneighborings_in_timed_region'), 'fallback', d.get('transport_fallback'))"; }
for n in 2 4; do
  MDP_RCCL_LIBRARY=$DOUBLE MDP_FAKE_RCCL_TIMEOUT_S=120 timeout -k 10 600 python3 bench.py --gpus $n --replicate 12 12 12 --temp 300 --steps 60 --warmup 5 --no-cpu-baseline > $O/double_rebomos_$n.json 2> $O/double_rebomos_$n.err
  echo "double rebomos --gpus $n rc=$?"; show $O/double_rebomos_$n.json
  MDP_RCCL_LIBRARY=$DOUBLE MDP_FAKE_RCCL_TIMEOUT_S=120 timeout -k 10 600 python3 bench.py --gpus $n --workload aeam --replicate 40 40 40 --temp 863 --steps 60 --warmup 5 --no-cpu-baseline > $O/double_aeam_$n.json 2> $O/double_aeam_$n.err
  echo "double aeam --gpus $n rc=$?"; show $O/double_aeam_$n.json
done
run1() { tag=$1; shift
  timeout -k 10 400 python3 bench.py --gpus 1 "$@" --no-cpu-baseline --no-host-mode --no-secondary > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$?"; show $O/$tag.json; }
self() { tag=$1; port=$2; shift 2
  MDP_BENCH_SELF_REMOTE=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 1 "$@" --no-cpu-baseline > $O/$tag.json 2> $O/$tag.err
  echo "$tag rc=$?"; show $O/$tag.json; grep -h "overlap policy" $O/$tag.err; }
run1 rebo_plain --replicate 12 12 12 --temp 300 --steps 200 --warmup 10
self self_rebo 29541 --replicate 12 12 12 --temp 300 --steps 200 --warmup 10
for p in split lead blocking first inline; do MDP_OVERLAP_POLICY=$p self self_rebo_$p 29543 --replicate 12 12 12 --temp 300 --steps 200 --warmup 10; done
run1 aeam_plain --workload aeam --replicate 63 63 63 --temp 863 --steps 200 --warmup 10
self self_aeam 29542 --workload aeam --replicate 63 63 63 --temp 863 --steps 200 --warmup 10
for p in split lead blocking inline; do MDP_OVERLAP_POLICY=$p self self_aeam_$p 29544 --workload aeam --replicate 63 63 63 --temp 863 --steps 200 --warmup 10; done
