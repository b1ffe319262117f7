#!/bin/bash
# multi-rank rehearsal on ONE GPU through the PLAIN command line `python3 bench.py --gpus N` (bench.py starts its own
# rank processes); MDP_BENCH_BACKEND=gloo: the ranks share the card and the halo is staged through the host.
# usage: profiles/rehearse.sh <tag> ; output under gpurun_out/<tag>/
set -u
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-rehearse}; mkdir -p $O
run() { tag=$1; n=$2; shift 2
  MDP_BENCH_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus $n "$@" --no-host-mode > $O/$tag.json 2> $O/$tag.err
  echo "$tag rc=$?"; grep '^{' $O/$tag.json | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); c=d['config']; print('  ', d['value'], d['ms_per_step'], 'n_gpus', d['n_gpus'], 'ranks', c.get('rccl_ranks'), c['parallelism'], c.get('pe_per_atom_end_eV'), c.get('temp_end_K'), 'reneigh', c.get('reneighborings_in_timed_region'), 'nlocal', c.get('nlocal_per_rank'), 'cpu', d.get('cpu_baseline',{}).get('value'), 'frac', d['roofline']['frac'])"
}
run rebo1 1 --replicate 10 10 10 --temp 300 --steps 40 --warmup 5 --no-secondary
run rebo2 2 --replicate 10 10 10 --temp 300 --steps 40 --warmup 5
run rebo4 4 --replicate 10 10 10 --temp 300 --steps 40 --warmup 5 --no-cpu-baseline
run aeam1 1 --workload aeam --replicate 30 30 30 --temp 863 --steps 60 --warmup 5 --no-cpu-baseline
run aeam4 4 --workload aeam --replicate 30 30 30 --temp 863 --steps 60 --warmup 5 --no-cpu-baseline
# BASELINE.json configs[4] (16.1 M atoms) on 1, 2 and 4 bricks: the same trajectory (PE/atom and T after the same
# number of steps), every multi-rank step on the phased path (exchanges behind the interior tiles)
if [ "${MDP_REHEARSE_16M:-1}" != "0" ]; then
for n in 1 2 4; do
  run aeam16m_$n $n --workload aeam --replicate 159 159 159 --temp 863 --steps 40 --warmup 5 --no-cpu-baseline --no-secondary
  python3 -c "
import json;d=json.load(open('$O/aeam16m_$n.json'));c=d['config'];print('   phased steps', c.get('aeam_steps_with_exchanges_behind_interior_tiles_rank0'), 'interior tiles', c.get('aeam_interior_tiles_rank0'), 'of', c.get('aeam_tiles_rank0'), 'ghost forces', c.get('aeam_ghost_force_exchange'))"
done
fi
# what must NOT produce a result: more RCCL ranks than GPUs
timeout -k 10 200 python3 bench.py --gpus 2 --replicate 4 4 4 --steps 2 --warmup 1 --no-cpu-baseline > $O/refuse.json 2> $O/refuse.err; echo "refuse rc=$? (non-zero expected), stdout bytes: $(wc -c < $O/refuse.json)"
# the N>1 code path of bench.py over RCCL itself with ONE rank: every periodic self-image travels through the transport
# to the rank itself (torch.distributed all_to_all on the "nccl" backend, then the library's ncclSend/ncclRecv)
for tr in torch native; do
  MDP_BENCH_SELF_REMOTE=1 MDP_BENCH_TRANSPORT=$tr timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --replicate 12 12 12 --temp 300 --steps 200 --warmup 10 --no-cpu-baseline > $O/self_$tr.json 2> $O/self_$tr.err
  echo "self_$tr rc=$? stdout lines: $(wc -l < $O/self_$tr.json)"; python3 -c "
import json;d=json.load(open('$O/self_$tr.json'));c=d['config'];print('  ', d['value'], d['ms_per_step'], c['transport'], c['pe_per_atom_end_eV'], c['temp_end_K'])"
  # aeam: position exchange, fp forward and ghost-force reverse exchanges through RCCL to the rank itself, against the
  # plain one-GPU run of the same system (aeam_plain below)
  MDP_BENCH_SELF_REMOTE=1 MDP_BENCH_TRANSPORT=$tr timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 1 --workload aeam --replicate 63 63 63 --temp 863 --steps 200 --warmup 10 --no-cpu-baseline > $O/self_aeam_$tr.json 2> $O/self_aeam_$tr.err
  echo "self_aeam_$tr rc=$? stdout lines: $(wc -l < $O/self_aeam_$tr.json)"; python3 -c "
import json;d=json.load(open('$O/self_aeam_$tr.json'));c=d['config'];print('  ', d['value'], d['ms_per_step'], c['transport'], c['pe_per_atom_end_eV'], c['temp_end_K'], 'phased steps', c.get('aeam_steps_with_exchanges_behind_interior_tiles_rank0'), 'interior', c.get('aeam_interior_tiles_rank0'), 'of', c.get('aeam_tiles_rank0'))"
done
run aeam_plain 1 --workload aeam --replicate 63 63 63 --temp 863 --steps 200 --warmup 10 --no-cpu-baseline --no-secondary
# ... and the plain one-GPU run of the REBO-MoS system of the self_* lines above (the 497 664-atom sub-domain of the 8-GPU
# headline run): self_native - rebo_plain = what the N>1 code path costs per step on top of the kernels
run rebo_plain 1 --replicate 12 12 12 --temp 300 --steps 200 --warmup 10 --no-cpu-baseline --no-secondary
python3 - $O <<'PY'
import json, sys
o = sys.argv[1]
def ms(tag):
    try:
        return json.loads(open(f"{o}/{tag}.json").read().strip().splitlines()[-1])["ms_per_step"]
    except Exception:
        return None
p, a = ms("rebo_plain"), ms("aeam_plain")
for tr in ("torch", "native"):
    r, q = ms(f"self_{tr}"), ms(f"self_aeam_{tr}")
    if p and r: print(f"rebomos 12x12x12 300 K: plain {p:.4f} ms, self-remote through {tr} {r:.4f} ms: +{(r - p) * 1e3:.1f} us per step")
    if a and q: print(f"aeam 63x63x63 863 K: plain {a:.4f} ms, self-remote through {tr} {q:.4f} ms: +{(q - a) * 1e3:.1f} us per step")
PY

