#!/usr/bin/env python3
"""bench.py -- headline benchmark: REBO-MoS bulk (in.rebomos-bulk cell replicated 24x24x24 =
3,981,312 atoms, BASELINE.json configs[3] / SURVEY.md 8d config #4) as a device-resident NVE run.

One "step" = one velocity-Verlet step around one pass of the hot path (PairREBOMoS::compute:
REBO centre kernels + LJ/gather kernel; --workload aeam: PairAEAM::compute passes 1-3) over all atoms of
the job; positions and lists are resident in HBM when the timed region starts.  As in the reference
inputs the timed region tallies energy and virial every `thermo` steps (10, in.rebomos-bulk:31; 100,
sample.in:28) and honours `neigh_modify every 1 check yes`: one GPU reads a deferred on-device
displacement flag every step, several GPUs agree on it every --check-every steps; reneighboring
(remap, migration between bricks, ghost derivation, list build) happens on the GPUs inside the timed
region.  N>1: one brick of the box per GPU (strong scaling, fixed total size), one ghost-position
all-to-all per step on RCCL, overlapped with the interior Lennard-Jones work.

Prints ONE JSON line on rank 0 (contract in the task statement)."""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from lammps_plugins_amd.host import capi, resident, system as S  # noqa: E402

POT_REBOMOS = os.path.join(ROOT, "tests", "golden", "potentials", "MoS.REBO.set5b")
POT_AEAM = os.path.join(ROOT, "tests", "golden", "potentials", "AlSi.aeam")

# algorithmic HBM bytes and FP64 flop-equivalents per atom-step
B_ALG = {"rebomos": 2040.0, "aeam": 400.0}
FLOP_ALG = {"rebomos": 25.0e3, "aeam": 7.0e3}
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6  # MI355X FP64 vector rate (public spec)
THERMO_EVERY = {"rebomos": 10, "aeam": 100}


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


# ------------------------------------------------------------------------------------------------ CPU baseline
def _cpu_sample(workload):
    """a bounded sample of the workload for the CPU oracle: (callable, atoms per call, description)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_bindings as ob
    import mdref
    orc = ob.load()
    if workload == "rebomos":
        P = orc.rebomos_params(POT_REBOMOS)
        s = S.replicate(S.rebomos_bulk_cell(), (3, 3, 2))
        eng = mdref.RebomosCPU(orc, P, s)
        what = "in.rebomos-bulk cell replicated 3x3x2 = %d atoms, force-only compute() calls" % s.n
        call = lambda: eng.orc.rebomos_compute(P, eng.nlocal, eng.x_all, eng.elem, eng.tag_all, eng.nn, eng.off,
                                               eng.nb, eflag=0, vflag=0)
    else:
        T = orc.aeam_pot(POT_AEAM)
        s = S.fcc_cell(4.045, 14, frac_type2=0.0075, seed=7683797)
        eng = mdref.AeamCPU(orc, T, s)
        what = "fcc 14x14x14 cells = %d atoms (0.75%% Si), force-only compute() calls" % s.n
        call = lambda: eng.orc.aeam_compute(T, eng.nlocal, eng.x_all, eng.type_all, eng.nn, eng.off, eng.nb,
                                            eflag=0, vflag=0)
    return call, s.n, what


def _cpu_worker(arg):
    workload, seconds = arg
    call, natoms, what = _cpu_sample(workload)
    call()
    t0 = time.perf_counter()
    n = 0
    while True:
        call()
        n += 1
        if time.perf_counter() - t0 > seconds or n >= 400:
            break
    return natoms * n, time.perf_counter() - t0, what, n


def cpu_baseline(workload, seconds=8.0):
    """the CPU oracle (port of the reference algorithm, oracle/) on a bounded sample of the same workload: one core
    (mirrors `1 MPI task x 1 thread`, log.rebomos-bulk.1:59) and all cores of this box's CPU share, every core
    working on its own replica of the sample (= ideal scaling of a spatial decomposition, log.rebomos-bulk.4:59).
    Only the pair computation is timed (99.7 % of the reference's loop, log.rebomos-bulk.1:65)."""
    work, dt, what, n = _cpu_worker((workload, seconds))
    one = dict(value=work / dt / 1e6, unit="Matom-steps/s", cores=1, kind="port",
               sample=what + ", %d calls in %.1f s" % (n, dt))
    import concurrent.futures as cf
    import multiprocessing as mp
    try:
        ncore = len(os.sched_getaffinity(0))
    except AttributeError:
        ncore = os.cpu_count() or 1
    ncore = max(1, min(ncore, 16))          # a one-GPU box's CPU share is 16 cores
    allc = None
    try:
        with cf.ProcessPoolExecutor(max_workers=ncore, mp_context=mp.get_context("spawn")) as ex:
            res = list(ex.map(_cpu_worker, [(workload, seconds)] * ncore))
        allc = dict(value=sum(r[0] for r in res) / max(r[1] for r in res) / 1e6, unit="Matom-steps/s", cores=ncore,
                    kind="port", sample=what + ", one replica per core, %d cores concurrently" % ncore)
    except Exception as e:  # noqa: BLE001 -- the all-core figure is informational
        log(f"[bench] all-core CPU baseline failed: {e}")
    return one, allc


# ------------------------------------------------------------------------------------------------ host mode
def host_mode_rate(s, p, skin, cutghost, steps=9):
    """PCIe-inclusive rate of the drop-in boundary (what a LAMMPS Pair::compute() sees): per step x of owned+ghost
    atoms goes up (24 B/atom), forces of owned atoms come back (24 B/atom).  REBO-MoS only; never `value`."""
    xw = S.wrap(s.box, s.x)
    owner, shift = S.make_ghosts(s.box, xw, cutghost)
    xa = np.ascontiguousarray(np.concatenate([xw, xw[owner] + shift @ s.box.h.T]))
    type_all = np.concatenate([s.type, s.type[owner]]).astype(np.int32)
    tag_all = np.concatenate([s.tag, s.tag[owner]]).astype(np.int32)
    n = s.n
    ctx = capi.Context(0)
    ctx.rebomos_set_params(p)
    ctx.set_atoms_host(n, xa, type_all, tag_all, 2, map_=[0, 0, 1])
    ctx.set_skin(skin)
    f = np.zeros((n, 3))
    eng, vir = capi.C.c_double(0.0), np.zeros(6)

    def compute():
        ctx._ck(ctx.L.mdp_rebomos_compute_host(ctx.h, 0, 0, capi._dp(f), capi.C.byref(eng), capi._dp(vir), None, None))

    compute()
    for _ in range(2):
        ctx.set_positions_host(xa)
        compute()
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        ctx.set_positions_host(xa)
        compute()
        times.append(time.perf_counter() - t0)
    ctx.close()
    return float(np.median(times)) * 1e3  # (median: the host threads of a freshly started box take a few steps to settle)


def kernel_source_sha():
    """hash of the sources that decide what the path kernels read and write (kernels, list builders, atom order)"""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "lammps-plugins_amd", "csrc")
    for name in ("aeam.hip", "domain.hip", "md.hip", "mdp_api.hip", "mdp_common.h", "rebomos.hip"):
        h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=["rebomos", "aeam"], default="rebomos")
    ap.add_argument("--replicate", type=int, nargs=3, default=None)
    ap.add_argument("--temp", type=float, default=0.0, help="initial temperature (in.rebomos-bulk: 0 K; sample.in: 863 K)")
    ap.add_argument("--check-every", type=int, default=10,
                    help="several GPUs: steps between the (collective) displacement checks; one GPU checks every step")
    ap.add_argument("--thermo", type=int, default=None, help="steps between energy/virial tallies (default: the input deck's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-mode", action="store_true")
    ap.add_argument("--inner-skin", type=float, default=None,
                    help="skin of the style's own device-built lists in A (default: library default 1.0, capped by "
                         "the host skin); they are rebuilt on the device when an atom has moved half of it")
    args = ap.parse_args()
    if args.replicate is None:
        args.replicate = [24, 24, 24] if args.workload == "rebomos" else [63, 63, 63]
    thermo_every = THERMO_EVERY[args.workload] if args.thermo is None else args.thermo

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
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if stage_host:
            dist.init_process_group(backend, rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    t_setup = time.perf_counter()
    s, wname = build_system(args)
    v0 = S.gaussian_velocities(s, args.temp, seed=1082337) if args.temp > 0 else None
    ctx = capi.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    p = None
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
    # transport of the bricks' exchanges: "torch" = all-to-all through torch.distributed (RCCL), the default;
    # MDP_BENCH_TRANSPORT=native = grouped ncclSend/ncclRecv inside libmdpair_hip.so (csrc/comm_rccl.hip; the
    # process group then only distributes the communicator id)
    native = dist is not None and os.environ.get("MDP_BENCH_TRANSPORT", "torch") == "native" and not stage_host
    if dist is None:
        tr = None
    elif native:
        def bcast(b):
            box = [b]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        tr = resident.NativeTransport(world, rank, bcast)
    else:
        tr = resident.Transport(dist, dev, stage_host)
    dom = resident.DeviceDomain(ctx, style, s, cutghost, skin, map_, v0=v0, transport=tr)
    dom.compute(1, 1)
    th = dom.thermo()
    pe0 = th["pe"]
    stats = ctx.md_neighbor_stats()
    if rank == 0:
        log(f"[bench] {wname}: {s.n} atoms, {world} GPU(s); rank0 nlocal={dom.nlocal} ghosts={dom.nself}+{dom.nrecv} "
            f"list/atom={stats[0] / max(dom.nlocal, 1):.1f} PE/atom={pe0 / s.n:.6f} eV "
            f"setup {time.perf_counter() - t_setup:.1f}s")

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # several GPUs check the displacement every --check-every steps: trigger early enough for that interval
    margin = 0.03 * max(args.check_every, 1)

    def run(nsteps, step0):
        """Verlet loop (Verlet::run): initial_integrate, [reneighbor when `check yes` fires], halo, force (energy /
        virial every `thermo` steps), final_integrate.  Returns the reneighborings it did."""
        rebuilds = 0
        for k in range(1, nsteps + 1):
            n = step0 + k
            ev = 1 if thermo_every and n % thermo_every == 0 else 0
            if dist is None:
                rebuild = "auto"
            else:
                rebuild = bool(args.check_every and n % args.check_every == 0 and dom.needs_rebuild(margin))
            b0 = dom.builds
            dom.step(ev, ev, rebuild=rebuild)
            rebuilds += dom.builds - b0
        return rebuilds

    run(args.warmup, 0)
    sync_all()
    style_builds0 = ctx.md_neighbor_stats()[7]
    prune0 = ctx.md_prune_stats()
    t0 = time.perf_counter()
    rebuilds = run(args.steps, args.warmup)
    sync_all()
    elapsed = time.perf_counter() - t0
    style_builds = ctx.md_neighbor_stats()[7] - style_builds0 if args.workload == "rebomos" else 0
    prune1 = ctx.md_prune_stats()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if stage_host else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel time of the hot path: HIP events on the compute stream, force-only steps (separate pass) ----
    ctx.set_timing(True)
    kms = np.zeros(8)
    nmeas = 20  # (long enough for a row pruning to weigh what it weighs in the run)
    for _ in range(nmeas):
        dom.step(0, 0)
        kms += np.array(ctx.get_timing())
    kms /= nmeas
    ctx.set_timing(False)

    # ---- one forced reneighboring (remap, migration, ghosts, lists; collective), wall time incl. its host syncs ----
    sync_all()
    t0 = time.perf_counter()
    dom.reneighbor()
    sync_all()
    reneighbor_ms = (time.perf_counter() - t0) * 1e3

    dom.compute(1, 1)
    th1 = dom.thermo()

    value = s.n * args.steps / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3
    if args.workload == "rebomos":
        lj = "rebo_lj_gather_kernel<16>" if os.environ.get("MDP_LJ_TILE", "1") == "0" else "rebo_lj_tile_kernel"
        knames = ["rebo_centre_kernel<4|8|12|16|32>", lj]
    else:
        # timed phases of the AEAM path (each is the named kernels back to back on the compute stream)
        dens = "aeam_tile_density_kernel" if os.environ.get("MDP_AEAM_PERSIST", "") == "0" else "aeam_ptile_kernel<density>"
        knames = [dens + "+aeam_density_ang_kernel", "aeam_embed_kernel",
                  "aeam_tile_force_kernel+aeam_force_ang_kernel"]
    kdom = int(np.argmax(kms[:len(knames)]))
    # algorithmic bytes of ONE pass of the path over this rank's atoms: SURVEY 8(d) per-atom figure x atoms.
    # `achieved` / `frac` use the time of ALL kernels of the path (the contract figure is per atom-step of the
    # whole compute(), not of its longest kernel); the dominant kernel alone is reported beside it.
    alg_bytes = B_ALG[args.workload] * dom.nlocal
    kall = float(kms[:len(knames)].sum())
    path_achieved = alg_bytes / (kall * 1e-3) / 1e9 if kall > 0 else 0.0
    kernel_achieved = alg_bytes / (kms[kdom] * 1e-3) / 1e9 if kms[kdom] > 0 else 0.0
    step_achieved = B_ALG[args.workload] * s.n / world / (ms_per_step * 1e-3) / 1e9
    flops_path = FLOP_ALG[args.workload] * dom.nlocal / (kall * 1e-3) / 1e12 if kall > 0 else 0.0
    flops_step = FLOP_ALG[args.workload] * s.n / world / (ms_per_step * 1e-3) / 1e12
    # HBM traffic from the PMC counters: collected by profiles/pmc_passes.sh in separate rocprofv3 runs and stored
    # with the hash of the kernel sources it was measured on; a stale entry is not reported
    traffic, traffic_note = None, "no PMC entry for this configuration"
    pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_file):
        try:
            tab = json.load(open(pmc_file))
            key = f"{args.workload}:{'x'.join(map(str, args.replicate))}:{world}"
            ent = tab.get(key)
            if isinstance(ent, dict):
                if ent.get("kernel_source_sha") == kernel_source_sha():
                    traffic, traffic_note = ent.get("bytes_per_step"), ent.get("note", "FETCH_SIZE x2 + WRITE_SIZE, path kernels of one step")
                else:
                    traffic_note = "PMC entry is stale (kernel sources changed since it was measured)"
        except Exception:  # noqa: BLE001
            pass
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
        "config": {"workload": wname, "atoms": s.n, "style": args.workload, "parallelism": f"spatial-dd{world}",
                   "transport": ("rccl (library, ncclSend/ncclRecv)" if native else "rccl (torch.distributed all_to_all)")
                   if not stage_host else backend + "-staged (rehearsal)",
                   "initial_temp_K": args.temp, "skin": skin, "thermo_every": thermo_every,
                   "displacement_check": "every step, deferred on-device flag" if dist is None
                   else f"every {args.check_every} steps, collective",
                   "reneighborings_in_timed_region": rebuilds, "reneighbor_wall_ms": round(reneighbor_ms, 3),
                   "inner_skin": (float(os.environ["MDP_INNER_SKIN"]) if "MDP_INNER_SKIN" in os.environ
                                  else "adaptive from 1.0") if args.workload == "rebomos" else None,
                   "style_list_builds_in_timed_region_rank0": int(style_builds),
                   "row_prunings_in_timed_region_rank0": prune1["prunings"] - prune0["prunings"],
                   "row_prunings_late_rank0": prune1["late"] - prune0["late"],
                   "row_pruning": (f"tile rows re-filtered to window + {prune1['buffer']:.2f} A from the current positions, "
                                   "own displacement trigger") if prune1["active"] else "off",
                   "pe_per_atom_start_eV": round(pe0 / s.n, 6), "pe_per_atom_end_eV": round(th1["pe"] / s.n, 6),
                   "temp_end_K": round(th1["temp"], 2)},
        "roofline": {"bound": "hbm", "achieved": round(path_achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(path_achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_note": traffic_note,
                     "basis": "algorithmic bytes of one compute() pass / time of ALL kernels of the path "
                              "(force-only step, HIP events on the compute stream)",
                     "algorithmic_bytes_per_pass": alg_bytes, "path_ms": round(kall, 4),
                     "all_kernels_ms": {n: round(float(m), 4) for n, m in zip(knames, kms)},
                     "dominant_kernel": {"name": knames[kdom], "ms": round(float(kms[kdom]), 4),
                                         "achieved": round(kernel_achieved, 2),
                                         "frac": round(kernel_achieved / HBM_PEAK_GBPS, 5)},
                     "whole_step": {"achieved": round(step_achieved, 2), "frac": round(step_achieved / HBM_PEAK_GBPS, 5)},
                     "fp64": {"bound": "fp64-valu", "flop_per_atom_step": FLOP_ALG[args.workload],
                              "achieved": round(flops_path, 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": round(flops_path / FP64_PEAK_TFLOPS, 5),
                              "whole_step_frac": round(flops_step / FP64_PEAK_TFLOPS, 5)}},
    }
    ctx.close()
    if rank == 0 and world == 1 and args.workload == "rebomos" and not args.no_host_mode:
        try:
            hm = host_mode_rate(s, p, skin, cutghost)
            out["host_mode_ms_per_step"] = round(hm, 3)
            out["host_mode_Matom_steps_per_s"] = round(s.n / hm / 1e3, 2)
        except Exception as e:  # noqa: BLE001 -- informational figure
            log(f"[bench] host-mode measurement failed: {e}")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        one, allc = cpu_baseline(args.workload)
        one["value"] = round(one["value"], 5)
        out["cpu_baseline"] = one
        out["gpu_over_cpu_1core"] = round(value / one["value"], 1)
        if allc is not None:
            allc["value"] = round(allc["value"], 5)
            out["cpu_baseline_allcores"] = allc
            out["gpu_over_cpu_allcores"] = round(value / allc["value"], 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
