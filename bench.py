#!/usr/bin/env python3
"""bench.py -- headline benchmark: REBO-MoS bulk (in.rebomos-bulk cell replicated 24x24x24 =
3,981,312 atoms, BASELINE.json configs[3] / SURVEY.md 8d config #4) as a device-resident NVE run.

One "step" = one velocity-Verlet step around one pass of the hot path (PairREBOMoS::compute:
REBO centre kernels + LJ/gather kernel) over all atoms of the job; positions and the neighbor list
are resident in HBM when the timed region starts.  `neigh_modify every 1 delay 0 check yes` is
honoured with the displacement check every 10 steps (rebuild + repack inside the timed region
when it fires).  N>1: atoms are spatially decomposed over N GPUs (strong scaling, fixed total
size) with one ghost-position all_to_all per step on RCCL.

Prints ONE JSON line on rank 0 (contract in the task statement)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from lammps_plugins_amd.host import capi, decomp, resident, system as S  # noqa: E402

POT_REBOMOS = os.path.join(ROOT, "tests", "golden", "potentials", "MoS.REBO.set5b")
POT_AEAM = os.path.join(ROOT, "tests", "golden", "potentials", "AlSi.aeam")

# algorithmic HBM bytes per atom-step
B_ALG = {"rebomos": 2040.0, "aeam": 400.0}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_system(args):
    if args.workload == "rebomos":
        s = S.replicate(S.rebomos_bulk_cell(), tuple(args.replicate))
        name = "REBO-MoS bulk: in.rebomos-bulk cell replicated %dx%dx%d" % tuple(args.replicate)
    else:
        s = S.fcc_cell(4.045, tuple(args.replicate), frac_type2=0.0075, seed=7683797)
        name = "AEAM AlSi: fcc a=4.045 %dx%dx%d cells, 0.75%% Si" % tuple(args.replicate)
    return s, name


def cpu_baseline(workload, seconds=12.0):
    """the CPU oracle (port of the reference algorithm, oracle/) on a bounded sample of the same
    workload, one core.  Only the pair computation is timed (99.7% of the reference's loop)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_bindings as ob
    import mdref
    orc = ob.load()
    if workload == "rebomos":
        P = orc.rebomos_params(POT_REBOMOS)
        s = S.replicate(S.rebomos_bulk_cell(), (3, 3, 2))
        eng = mdref.RebomosCPU(orc, P, s)
        sample = "in.rebomos-bulk cell replicated 3x3x2 = %d atoms, force-only compute() calls" % s.n
        call = lambda: eng.orc.rebomos_compute(P, eng.nlocal, eng.x_all, eng.elem, eng.tag_all, eng.nn, eng.off,
                                               eng.nb, eflag=0, vflag=0)
    else:
        T = orc.aeam_pot(POT_AEAM)
        s = S.fcc_cell(4.045, 14, frac_type2=0.0075, seed=7683797)
        eng = mdref.AeamCPU(orc, T, s)
        sample = "fcc 14x14x14 cells = %d atoms (0.75%% Si), force-only compute() calls" % s.n
        call = lambda: eng.orc.aeam_compute(T, eng.nlocal, eng.x_all, eng.type_all, eng.nn, eng.off, eng.nb,
                                            eflag=0, vflag=0)
    call()
    t0 = time.perf_counter()
    n = 0
    while True:
        call()
        n += 1
        if time.perf_counter() - t0 > seconds or n >= 400:
            break
    dt = time.perf_counter() - t0
    return dict(value=s.n * n / dt / 1e6, unit="Matom-steps/s", cores=1, kind="port",
                sample=sample + ", %d calls in %.1f s" % (n, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=["rebomos", "aeam"], default="rebomos")
    ap.add_argument("--replicate", type=int, nargs=3, default=None)
    ap.add_argument("--temp", type=float, default=0.0, help="initial temperature (in.rebomos-bulk: 0 K)")
    ap.add_argument("--check-every", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inner-skin", type=float, default=None,
                    help="skin of the style's own device-built lists in A (default: library default 1.0, capped by "
                         "the host skin); they are rebuilt on the device when an atom has moved half of it")
    args = ap.parse_args()
    if args.replicate is None:
        args.replicate = [24, 24, 24] if args.workload == "rebomos" else [63, 63, 63]

    if args.inner_skin is not None:
        os.environ["MDP_INNER_SKIN"] = str(args.inner_skin)
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"[bench] note: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # MDP_BENCH_BACKEND=gloo is a rehearsal switch for boxes with fewer GPUs than ranks: every rank uses
    # GPU (local_rank mod #GPUs) and the halo is staged through the host.  The judged runs use RCCL.
    backend = os.environ.get("MDP_BENCH_BACKEND", "nccl")
    stage_host = backend != "nccl"
    if stage_host:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if stage_host:
            dist.init_process_group(backend, rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    t_setup = time.perf_counter()
    s, wname = build_system(args)
    v0 = S.gaussian_velocities(s, args.temp, seed=1082337) if args.temp > 0 else None
    ctx = capi.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    if args.workload == "rebomos":
        p = capi.read_rebomos_file(POT_REBOMOS)
        ctx.rebomos_set_params(p)
        style, skin, map_ = capi.STYLE_REBOMOS, 2.0, [0, 0, 1]
        cutghost = 3.0 * p.rcmax[0][0] + skin
    else:
        af = capi.AeamFile(POT_AEAM)
        tabs = af.build()
        ctx.aeam_set_tables(tabs)
        style, skin, map_ = capi.STYLE_AEAM, 1.0, None
        cutghost = float(af.cut_table(tabs).max()) + skin
        s.mass[1:3] = af.mass[:2]
    dev = torch.device("cuda", local_rank)
    dom = resident.make_domain(ctx, style, s, cutghost, skin, map_, v0=v0, dist=dist, device=dev,
                               stage_host=stage_host and dist is not None)
    dom.build_neighbors()
    dom.compute(eflag=1, vflag=1)
    th = dom.thermo()
    pe0 = th["pe"]
    rdev = "cpu" if stage_host else "cuda"   # device of the small reduction tensors
    if dist is not None:
        t = torch.tensor([th["pe"]], dtype=torch.float64, device=rdev)
        dist.all_reduce(t)
        pe0 = float(t.item())
    stats = ctx.md_neighbor_stats()
    if rank == 0:
        log(f"[bench] {wname}: {s.n} atoms, {world} GPU(s); rank0 nlocal={dom.nlocal} nghost={dom.nghost} "
            f"master-list/atom={stats[0] / max(dom.nlocal, 1):.1f} PE/atom={pe0 / s.n:.6f} eV "
            f"setup {time.perf_counter() - t_setup:.1f}s")

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run(nsteps, step0):
        """Verlet loop.  Every --check-every steps: `neigh_modify check yes` on the host's skin (max over
        ranks); when it fires, atoms are re-wrapped / re-assigned and ghosts re-derived (Comm::exchange +
        borders) before the step.  Multi-GPU REBO-MoS steps overlap the ghost exchange with the interior
        Lennard-Jones work (RankDomain.step_overlapped)."""
        nonlocal dom
        rebuilds = 0
        for k in range(1, nsteps + 1):
            if args.check_every and (step0 + k) % args.check_every == 0:
                need = dom.needs_rebuild()
                if dist is not None:
                    t = torch.tensor([1.0 if need else 0.0], device=rdev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    need = bool(t.item() > 0)
                if need:
                    dom = resident.reneighbor(dom, s, cutghost, map_, dist=dist, device=dev)
                    dom.compute(0, 0)      # forces of the current positions for the next half kick
                    rebuilds += 1
            dom.step(0, 0)
        return rebuilds

    run(args.warmup, 0)
    sync_all()
    style_builds0 = ctx.md_neighbor_stats()[7]
    t0 = time.perf_counter()
    rebuilds = run(args.steps, args.warmup)
    sync_all()
    elapsed = time.perf_counter() - t0
    style_builds = ctx.md_neighbor_stats()[7] - style_builds0 if args.workload == "rebomos" else 0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel time of the hot path, HIP events on the compute stream (separate pass) ----------
    ctx.set_timing(True)
    kms = np.zeros(8)
    nmeas = 5
    for _ in range(nmeas):
        dom.step(0, 0)
        kms += np.array(ctx.get_timing())
    kms /= nmeas
    ctx.set_timing(False)

    dom.compute(eflag=1, vflag=1)
    th1 = dom.thermo()

    value = s.n * args.steps / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3
    if args.workload == "rebomos":
        lj = "rebo_lj_gather_kernel<16>" if os.environ.get("MDP_LJ_TILE", "1") == "0" else "rebo_lj_tile_kernel"
        knames = ["rebo_centre_kernel<4|8|12|16|32>", lj]
    else:
        # timed phases of the AEAM path (each is the named kernels back to back on the compute stream)
        knames = ["aeam_tile_density_kernel+aeam_density_ang_kernel", "aeam_embed_kernel",
                  "aeam_tile_force_kernel+aeam_force_ang_kernel"]
    kdom = int(np.argmax(kms[:len(knames)]))
    # algorithmic bytes of ONE launch of the dominant kernel: SURVEY 8(d) per-atom figure x atoms per launch
    alg_bytes = B_ALG[args.workload] * dom.nlocal
    achieved = alg_bytes / (kms[kdom] * 1e-3) / 1e9 if kms[kdom] > 0 else 0.0
    # the same bytes over ALL kernels of the path (atom-steps/s x bytes) -- the stricter figure
    kall = float(kms[:len(knames)].sum())
    path_achieved = alg_bytes / (kall * 1e-3) / 1e9 if kall > 0 else 0.0
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_file):
        try:
            tab = json.load(open(pmc_file))
            key = f"{args.workload}:{'x'.join(map(str, args.replicate))}:{world}"
            traffic = tab.get(f"{key}:{knames[kdom]}", tab.get(key))
        except Exception:
            traffic = None
    out = {
        "metric": "Matom-steps/sec",
        "value": round(value, 4),
        "unit": "Matom-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "ns_per_day": round(args.steps / elapsed * 0.001 * 86.4, 4),
        "config": {"workload": wname, "atoms": s.n, "style": args.workload, "parallelism": f"spatial-dd{world}", "transport": "rccl" if not stage_host else backend + "-staged (rehearsal)",
                   "initial_temp_K": args.temp, "skin": skin, "neighbor_rebuilds_in_timed_region": rebuilds,
                   "inner_skin": (float(os.environ["MDP_INNER_SKIN"]) if "MDP_INNER_SKIN" in os.environ
                                  else "adaptive from 1.0") if args.workload == "rebomos" else None,
                   "style_list_builds_in_timed_region_rank0": int(style_builds),
                   "pe_per_atom_start_eV": round(pe0 / s.n, 6)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "kernel": knames[kdom],
                     "kernel_ms": round(float(kms[kdom]), 4),
                     "all_kernels_ms": {n: round(float(m), 4) for n, m in zip(knames, kms)},
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "path_ms": round(kall, 4), "path_achieved": round(path_achieved, 2),
                     "path_frac": round(path_achieved / HBM_PEAK_GBPS, 5)},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline(args.workload)
        cb["value"] = round(cb["value"], 5)
        out["cpu_baseline"] = cb
        out["gpu_over_cpu_1core"] = round(value / cb["value"], 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
