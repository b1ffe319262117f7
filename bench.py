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
all-to-all per step on RCCL, overlapped with the interior REBO centres.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own N rank processes
(python -m torch.distributed.run, fresh children, the parent never touches the GPU) and relays rank 0's line;
it exits non-zero -- never falls back to fewer ranks -- when a rank does not come up, when the box has fewer
GPUs than ranks, or when RCCL reports another rank count.

Prints ONE JSON line on rank 0 (contract in the task statement).  The default one-GPU run adds a `secondary`
block: the same REBO-MoS system from 300 K (list builds and row prunings inside the timed region) and the AEAM
configuration #3 (1,000,188 atoms, 863 K, 1000 steps, check every step), each with its own roofline / CPU baseline."""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

# algorithmic HBM bytes and FP64 flop-equivalents per atom-step
B_ALG = {"rebomos": 2040.0, "aeam": 400.0}
FLOP_ALG = {"rebomos": 25.0e3, "aeam": 7.0e3}
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6  # MI355X FP64 vector rate (public spec)
THERMO_EVERY = {"rebomos": 10, "aeam": 100}
DEFAULT_REPLICATE = {"rebomos": [24, 24, 24], "aeam": [63, 63, 63]}
POT_REBOMOS = os.path.join(ROOT, "tests", "golden", "potentials", "MoS.REBO.set5b")
POT_AEAM = os.path.join(ROOT, "tests", "golden", "potentials", "AlSi.aeam")


def _jsonable(o):
    """numpy scalars / arrays that found their way into the result (a line that cannot be printed is a lost run)"""
    if hasattr(o, "tolist"):
        return o.tolist()
    if hasattr(o, "item"):
        return o.item()
    return str(o)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------------ rank launcher
def _rank_tree(n: int, env: dict, deadline: float):
    """start n fresh rank processes of this script (own process group) and wait for them: (exit code, result line or
    None, what went wrong or None).  Whatever ends the parent -- a signal, the deadline, an exception -- ends the ranks
    too (first SIGTERM, then SIGKILL), so no rank is left holding a GPU."""
    import signal
    import threading
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"[bench] starting {n} ranks: {' '.join(cmd[1:8])} ...")
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, start_new_session=True)

    def stop_ranks(sig=signal.SIGTERM):
        try:
            os.killpg(p.pid, sig)
        except (ProcessLookupError, PermissionError):
            pass

    def on_signal(signum, frame):
        log(f"[bench] signal {signum}: stopping the ranks")
        stop_ranks()
        raise SystemExit(128 + signum)

    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    timed_out = []

    def on_deadline():
        timed_out.append(True)
        log(f"[bench] the ranks did not finish within {deadline:.0f} s: stopping them")
        stop_ranks()
        time.sleep(10)
        stop_ranks(signal.SIGKILL)

    timer = threading.Timer(deadline, on_deadline)
    timer.daemon = True
    timer.start()
    line = None
    try:
        for out in p.stdout:
            out = out.rstrip("\n")
            if out.startswith("{") and '"metric"' in out:
                line = out
            elif out:
                log(out)
        rc = p.wait()
    finally:
        timer.cancel()
        if p.poll() is None:
            stop_ranks()
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                stop_ranks(signal.SIGKILL)
                p.wait()
        for sg, h in old.items():
            signal.signal(sg, h)
    if timed_out:
        return 1, None, f"the ranks did not finish within {deadline:.0f} s"
    if rc != 0:
        return (rc if rc > 0 else 1), None, f"the rank processes failed (exit code {rc})"
    if line is None:
        return 1, None, "the ranks exited without a result line"
    try:
        got = json.loads(line)
    except ValueError:
        return 1, None, "rank 0 printed a damaged result line"
    if got.get("n_gpus") != n or got.get("config", {}).get("rccl_ranks", n) != n:
        return 1, None, f"the result is for {got.get('n_gpus')} ranks, {n} were asked for"
    return 0, line, None


def launch_ranks(n: int) -> int:
    """start n fresh rank processes of this script and relay rank 0's JSON line.  The parent makes no GPU call.
    The default transport is the library's own (csrc/comm_rccl.hip).  If those ranks fail or pass their deadline
    (MDP_BENCH_NATIVE_DEADLINE_S, 300 s), ONE fallback: fresh rank processes with MDP_BENCH_TRANSPORT=torch (all-to-all
    through torch.distributed), and the line says so in `transport_fallback` -- never silently.  Both failing: non-zero."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    total = float(os.environ.get("MDP_BENCH_DEADLINE_S", "3000"))
    native_first = env.get("MDP_BENCH_TRANSPORT", "native") == "native"
    t0 = time.time()
    first_deadline = min(total, float(os.environ.get("MDP_BENCH_NATIVE_DEADLINE_S", "300"))) if native_first else total
    rc, line, why = _rank_tree(n, env, first_deadline)
    if rc != 0 and native_first and os.environ.get("MDP_BENCH_NO_FALLBACK", "0") in ("", "0"):
        left = total - (time.time() - t0)
        log(f"[bench] library transport: {why}; starting fresh ranks on the torch.distributed transport ({left:.0f} s left)")
        if left > 30:
            env2 = dict(env, MDP_BENCH_TRANSPORT="torch", MDP_BENCH_FALLBACK_REASON="library transport (csrc/comm_rccl.hip): " + why)
            env2.pop("MDP_BENCH_TEST_FAIL_NATIVE", None)
            rc, line, why = _rank_tree(n, env2, left)
    if rc != 0:
        log(f"[bench] {why}; no result")
        return rc
    print(line, flush=True)
    return 0


# ------------------------------------------------------------------------------------------------ CPU baseline
def _cpu_sample(workload):
    """a bounded sample of the workload for the CPU oracle: (callable, atoms per call, description)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as graft
    graft.load_package()
    from lammps_plugins_amd.host import system as S
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
    one = dict(value=round(work / dt / 1e6, 5), unit="Matom-steps/s", cores=1, kind="port",
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
        allc = dict(value=round(sum(r[0] for r in res) / max(r[1] for r in res) / 1e6, 5), unit="Matom-steps/s",
                    cores=ncore, kind="port", sample=what + ", one replica per core, %d cores concurrently" % ncore)
    except Exception as e:  # noqa: BLE001 -- the all-core figure is informational
        log(f"[bench] all-core CPU baseline failed: {e}")
    return one, allc


def attach_cpu_baseline(out, workload, seconds):
    one, allc = cpu_baseline(workload, seconds)
    out["cpu_baseline"] = one
    out["gpu_over_cpu_1core"] = round(out["value"] / one["value"], 1)
    if allc is not None:
        out["cpu_baseline_allcores"] = allc
        out["gpu_over_cpu_allcores"] = round(out["value"] / allc["value"], 1)


# ------------------------------------------------------------------------------------------------ host mode
def host_mode_rate(E, s, workload, pot, skin, cutghost, steps=9):
    """PCIe-inclusive rate of the drop-in boundary (what a LAMMPS Pair::compute() on one rank sees): per step x of the
    owned atoms goes up (24 B/atom), their forces come back (24 B/atom).  Never `value`."""
    import numpy as np
    capi, S = E["capi"], E["S"]
    xw = S.wrap(s.box, s.x)
    owner, shift = S.make_ghosts(s.box, xw, cutghost)
    xa = np.ascontiguousarray(np.concatenate([xw, xw[owner] + S.mul_upper(shift, s.box.h)]))
    type_all = np.concatenate([s.type, s.type[owner]]).astype(np.int32)
    tag_all = np.concatenate([s.tag, s.tag[owner]]).astype(np.int32)
    n, nall = s.n, len(xa)
    ctx = capi.Context(0)
    eng, vir = capi.C.c_double(0.0), np.zeros(6)
    ctx.set_box_host(s.box)          # as the adapters do on one rank (plugin/pair_*.cpp): the images stay on the device
    if workload == "rebomos":
        ctx.rebomos_set_params(pot)
        ctx.set_atoms_host(n, xa, type_all, tag_all, 2, map_=[0, 0, 1])
        ctx.set_skin(skin)
        f = np.zeros((n, 3))

        def compute():
            ctx._ck(ctx.L.mdp_rebomos_compute_host(ctx.h, 0, 0, capi._dp(f), capi.C.byref(eng), capi._dp(vir), None, None))
    else:
        ctx.aeam_set_tables(pot)
        ctx.aeam_device_lists(True)
        ctx.set_atoms_host(n, xa, type_all, tag_all, 2)
        ctx.set_skin(skin)
        f = np.zeros((nall, 3))
        fp = np.zeros(nall)
        local_halo = ctx.host_ghosts_derived()

        def compute():
            # PairAEAM::compute in the adapter (plugin/pair_aeam.cpp): density half, the style's forward_comm of fp,
            # force half.  On one periodic rank fp and the images' share of the forces stay on the device; otherwise
            # the images copy their owners' fp on the host, as Comm::forward_comm does
            if local_halo:
                ctx._ck(ctx.L.mdp_aeam_density_host(ctx.h, 0, None, None, capi.C.byref(eng), None))
                ctx._ck(ctx.L.mdp_aeam_force_host(ctx.h, 0, 0, None, capi._dp(f), capi.C.byref(eng), capi._dp(vir), None, None))
                return
            ctx._ck(ctx.L.mdp_aeam_density_host(ctx.h, 0, capi._dp(fp), None, capi.C.byref(eng), None))
            fp[n:] = fp[owner]
            ctx._ck(ctx.L.mdp_aeam_force_host(ctx.h, 0, 0, capi._dp(fp), capi._dp(f), capi.C.byref(eng),
                                              capi._dp(vir), None, None))

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
    derived = ctx.host_ghosts_derived()
    ctx.close()
    # (median: the host threads of a freshly started box take a few steps to settle)
    return float(np.median(times)) * 1e3, derived


def kernel_source_sha():
    """hash of the sources that decide what the path kernels read and write (kernels, list builders, atom order)"""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "lammps-plugins_amd", "csrc")
    for name in ("aeam.hip", "domain.hip", "md.hip", "mdp_api.hip", "mdp_common.h", "rebomos.hip"):
        h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def pmc_entry(workload, replicate, world):
    """HBM traffic from the PMC counters: collected by profiles/pmc_passes.sh in separate rocprofv3 runs and stored
    with the hash of the kernel sources it was measured on; a stale entry is not reported"""
    pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(pmc_file):
        return None, "no PMC table in profiles/"
    try:
        tab = json.load(open(pmc_file))
    except ValueError:
        return None, "profiles/pmc_traffic.json is damaged"
    ent = tab.get(f"{workload}:{'x'.join(map(str, replicate))}:{world}")
    if not isinstance(ent, dict):
        return None, "no PMC entry for this configuration"
    if ent.get("kernel_source_sha") != kernel_source_sha():
        return None, "PMC entry is stale (kernel sources changed since it was measured)"
    return ent, "stored PMC figure (profiles/pmc_traffic.json, measured on these kernel sources; not re-measured in this run): " + \
        ent.get("note", "FETCH_SIZE x2 + WRITE_SIZE, path kernels of one step")



# ------------------------------------------------------------------------------------------------ binding resource
N_CU, N_XCD, N_SIMD = 256, 8, 1024
L1_LOOKUP_CEILING = 0.9    # profiles/ubench/tcp_gather: vector-L1 lookups per clock and CU, every lane another row, all hits


def binding_block(workload, rep, world, dominant_short):
    """What binds the dominant kernel when it is not HBM (it is not: traffic is below the algorithmic bytes for rebomos and the
    rate is a tenth of the roof for aeam): from the STORED counter file of the same configuration (profiles/pmc_binding.json,
    separate rocprofv3 --pmc passes, reported only while the hash of csrc/* matches the sources it was measured on).
      rebomos kernels: FP64 vector issue -- SQ_INSTS_VALU wave instructions x 4 cycles / (SIMDs x kernel cycles)
      aeam pair force: vector-L1 lookups per clock and CU -- TCP_TOTAL_CACHE_ACCESSES / (kernel cycles x CUs), against the
                       0.9 a CU sustains when every lane reads another row and everything hits (profiles/ubench/tcp_gather)
    kernel cycles = GRBM_GUI_ACTIVE / XCDs of the same launch, so the figure does not depend on this run's clock."""
    f = os.path.join(ROOT, "profiles", "pmc_binding.json")
    if not os.path.exists(f):
        return None
    try:
        ent = json.load(open(f)).get(f"{workload}:{'x'.join(map(str, rep))}:{world}")
    except ValueError:
        return None
    if not isinstance(ent, dict) or ent.get("kernel_source_sha") != kernel_source_sha():
        return {"resource": None, "note": "no counter entry for these kernel sources (profiles/pmc_binding.json is stale or has no such configuration)"}
    per = {}
    for name, c in ent.get("kernels", {}).items():
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / N_XCD
        if cyc <= 0:
            continue
        per[name] = {"fp64_valu_issue_frac": round(c.get("SQ_INSTS_VALU", 0.0) * 4.0 / (N_SIMD * cyc), 4),
                     "l1_lookups_per_clk_cu": round(c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / (cyc * N_CU), 4),
                     "lds_conflict_share_of_lds_cycles": (round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_ACTIVE_INST_LDS"], 3)
                                                          if c.get("SQ_ACTIVE_INST_LDS") else None),
                     "waves_waiting_share": (round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3) if c.get("SQ_WAVE_CYCLES") else None),
                     "kernel_cycles": int(cyc)}
    k = per.get(dominant_short)
    if not k:
        return {"resource": None, "note": f"no counters for {dominant_short}", "per_kernel": per}
    if workload == "aeam":
        out = {"resource": "vector-L1 lookups per clock and CU (per-lane gathers of spline rows: one lookup moves at most 16 B of a lane)",
               "achieved": k["l1_lookups_per_clk_cu"], "ceiling": L1_LOOKUP_CEILING, "unit": "lookups/clk/CU",
               "frac": round(k["l1_lookups_per_clk_cu"] / L1_LOOKUP_CEILING, 4)}
    else:
        out = {"resource": "FP64 vector issue (one wave instruction per SIMD every 4 cycles)",
               "achieved": k["fp64_valu_issue_frac"], "ceiling": 1.0, "unit": "share of issue slots",
               "frac": k["fp64_valu_issue_frac"]}
    out.update(kernel=dominant_short, per_kernel=per,
               source="stored counters (profiles/pmc_binding.json, rocprofv3 --pmc passes on these kernel sources; not re-measured "
                      "in this run): " + ent.get("note", ""))
    return out


# ------------------------------------------------------------------------------------------------ the drop-in path
def plugin_load_run(example, subs, timeout=900, np=1):
    """`minilmp -in <example>` as a FRESH child process (this process holds no context any more): the reference's plugin
    surface -- plugin load, pair_style, fix nve/mdp, run -- on the mini-host.  Returns ms per step from the host's own
    `Loop time` line, the fix's download count and the thermo rows."""
    pkg = os.path.join(ROOT, "lammps-plugins_amd")
    text = open(os.path.join(pkg, "examples", example)).read()
    for old, new in subs.items():
        if old not in text:
            raise RuntimeError(f"{example}: '{old}' not found")
        text = text.replace(old, new)
    p = subprocess.run([os.path.join(pkg, "minilmp")] + (["-np", str(np)] if np > 1 else []), input=text, capture_output=True,
                       text=True, cwd=pkg, timeout=timeout, env=dict(os.environ, MDP_FIX_STATS="1"))
    if p.returncode != 0:
        raise RuntimeError(f"minilmp failed ({p.returncode}): {p.stderr[-400:]}")
    import re
    m = re.search(r"Loop time of ([0-9.eE+-]+) on \d+ procs for (\d+) steps with (\d+) atoms", p.stdout)
    d = re.search(r"fix nve/mdp: (\d+) downloads", p.stdout)
    br = re.search(r"fix nve/mdp: (\d+) bricks, (\d+) reneighborings on the device, (\d+) returns of the atoms", p.stdout)
    rows, on = [], False
    for line in p.stdout.splitlines():
        w = line.split()
        if w[:1] == ["Step"]:
            on = True
        elif on and w and w[0].lstrip("-").isdigit():
            rows.append([float(x) for x in w])
        elif on and line.startswith("Loop time"):
            on = False
    nb = re.search(r"update: every = (\d+) steps, delay = (\d+) steps, check = (\w+)", p.stdout)
    builds = re.search(r"Neighbor list builds = (\d+)", p.stdout)
    return dict(ms_per_step=round(float(m.group(1)) / int(m.group(2)) * 1e3, 4), steps=int(m.group(2)), atoms=int(m.group(3)),
                Matom_steps_per_s=round(int(m.group(3)) * int(m.group(2)) / float(m.group(1)) / 1e6, 2),
                downloads=int(d.group(1)) if d else (int(br.group(3)) if br else None), thermo_rows=rows, ranks=np,
                bricks=int(br.group(1)) if br else None, device_reneighborings=int(br.group(2)) if br else None,
                neighbor_settings_in_effect=(f"every {nb.group(1)} delay {nb.group(2)} check {nb.group(3)}" if nb else None),
                host_neighbor_list_builds=int(builds.group(1)) if builds else None,
                check_yes_decided_on_device="check yes decided on the device" in p.stdout, input="examples/" + example)


def resident_thermo_rows(E, steps):
    """thermo rows (step, temp, press, pe, ke) of the headline system in a device-resident run, at the given steps"""
    capi, resident, S = E["capi"], E["resident"], E["S"]
    s = S.replicate(S.rebomos_bulk_cell(), tuple(DEFAULT_REPLICATE["rebomos"]))
    ctx = capi.Context(0)
    pot = capi.read_rebomos_file(POT_REBOMOS)
    ctx.rebomos_set_params(pot)
    dom = resident.DeviceDomain(ctx, capi.STYLE_REBOMOS, s, 3.0 * pot.rcmax[0][0] + 2.0, 2.0, [0, 0, 1])
    dom.compute(1, 1)
    rows, n = [], 0
    for target in steps:
        while n < target:
            n += 1
            ev = 1 if n == target else 0
            dom.step(ev, ev, rebuild="auto", defer_final=not ev)
        t = dom.thermo()
        rows.append([float(n), float(t["temp"]), float(t["press"]), float(t["pe"]), float(t["ke"])])
    ctx.close()
    return rows


# ------------------------------------------------------------------------------------------------ one measured job
def run_job(E, job, par):
    """job: workload, replicate, temp, steps, warmup, thermo_every, check_every, inner_skin
    par: world, rank, local_rank, dist, dev, stage_host, native.  Returns (result dict, pieces for the host-mode leg)."""
    import numpy as np
    import torch
    capi, resident, S = E["capi"], E["resident"], E["S"]
    world, rank, dist, dev = par["world"], par["rank"], par["dist"], par["dev"]
    stage_host, native = par["stage_host"], par["native"]
    wl, rep = job["workload"], job["replicate"]
    t_setup = time.perf_counter()
    dis = job.get("disorder")          # off-lattice variants of the secondary block (SURVEY Appendix C constructions)
    if wl == "rebomos":
        s = S.replicate(S.rebomos_bulk_cell(), tuple(rep))
        wname = "REBO-MoS bulk: in.rebomos-bulk cell replicated %dx%dx%d" % tuple(rep)
        if dis:
            s = S.jitter(S.scale(s, dis["scale"]), dis["jitter"], seed=dis["seed"])
            wname += ", box and coordinates x %.2f, uniform jitter +-%.2f A (R-strain-112)" % (dis["scale"], dis["jitter"])
    else:
        frac = dis["frac2"] if dis else 0.0075
        s = S.fcc_cell(4.045, tuple(rep), frac_type2=frac, seed=7683797)
        wname = "AEAM AlSi: fcc a=4.045 %dx%dx%d cells, %.2f%% Si" % (*rep, 100.0 * frac)
    v0 = S.gaussian_velocities(s, job["temp"], seed=1082337) if job["temp"] > 0 else None
    ctx = capi.Context(par["local_rank"])
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    if wl == "rebomos":
        pot = capi.read_rebomos_file(POT_REBOMOS)
        ctx.rebomos_set_params(pot)
        style, skin, map_ = capi.STYLE_REBOMOS, 2.0, [0, 0, 1]
        cutghost = 3.0 * pot.rcmax[0][0] + skin
    else:
        af = capi.AeamFile(POT_AEAM)
        pot = af.build()
        ctx.aeam_set_tables(pot)
        ctx._af = af
        style, skin, map_ = capi.STYLE_AEAM, 1.0, None     # sample.in:17
        cutghost = float(af.cut_table(pot).max()) + skin
        s.mass[1:3] = af.mass[:2]
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
    dom = resident.DeviceDomain(ctx, style, s, cutghost, skin, map_, v0=v0, transport=tr, self_remote=par.get("self_remote", False))
    dom.compute(1, 1)
    th = dom.thermo()
    pe0 = th["pe"]
    overlap = None
    if native:
        # part of the setup, before the W warm-up steps: the library tries its orders of compute against exchanges on
        # THIS machine and keeps the cheapest (mdp_dd_comm_step_info); a fixed policy (MDP_OVERLAP_POLICY) skips this
        overlap = dom.tune_overlap()
    stats = ctx.md_neighbor_stats()
    if rank == 0:
        log(f"[bench] {wname}: {s.n} atoms, {world} GPU(s), T0 {job['temp']} K; rank0 nlocal={dom.nlocal} "
            f"ghosts={dom.nself}+{dom.nrecv} PE/atom={pe0 / s.n:.6f} eV setup {time.perf_counter() - t_setup:.1f}s")

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    thermo_every, check_every = job["thermo_every"], job["check_every"]
    # several GPUs check the displacement every --check-every steps: trigger early enough for that interval
    margin = 0.03 * max(check_every, 1)

    def run(nsteps, step0, on_step=None):
        """Verlet loop (Verlet::run): initial_integrate, [reneighbor when `check yes` fires], halo, force (energy /
        virial every `thermo` steps), final_integrate.  Returns the reneighborings it did."""
        rebuilds = 0
        for k in range(1, nsteps + 1):
            n = step0 + k
            ev = 1 if thermo_every and n % thermo_every == 0 else 0
            if dist is None:
                rebuild = "auto"
            elif native:
                rebuild = "halo"      # the collective decision rides with the position exchange (mdp_dd_comm_step_begin)
            else:
                rebuild = bool(check_every and n % check_every == 0 and dom.needs_rebuild(margin))
            b0 = dom.builds
            # steps without thermo output leave their final_integrate to the first kernel of the next step (one pass
            # over the atoms for both half-kicks, same arithmetic); thermo steps complete theirs at once
            dom.step(ev, ev, rebuild=rebuild, defer_final=not ev and k < nsteps)
            rebuilds += dom.builds - b0
            if on_step is not None:
                on_step(dom.builds - b0, ev)
        return rebuilds

    run(job["warmup"], 0)
    sync_all()
    style_builds0 = ctx.md_neighbor_stats()[7]
    prune0 = ctx.md_prune_stats()
    t0 = time.perf_counter()
    rebuilds = run(job["steps"], job["warmup"])
    sync_all()
    elapsed = time.perf_counter() - t0
    style_builds = ctx.md_neighbor_stats()[7] - style_builds0 if wl == "rebomos" else 0
    lstate = ctx.md_list_state()
    prune1 = ctx.md_prune_stats()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if stage_host else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel time of the hot path: HIP events on the compute stream, force-only steps of the SAME Verlet loop
    # (same rebuild policy; steps that reneighbored are left out of the average) ----
    ctx.set_timing(True)
    kms, kcount = np.zeros(8), [0]

    def collect(rebuilt, ev):
        if not rebuilt and not ev:
            kms[:] += np.array(ctx.get_timing())
            kcount[0] += 1

    nmeas = 20  # (long enough for a row pruning to weigh what it weighs in the run)
    run(nmeas, job["warmup"] + job["steps"], on_step=collect)   # (thermo steps fall where they fall and are left out)
    kms /= max(kcount[0], 1)
    ctx.set_timing(False)

    # ---- one forced reneighboring (remap, migration, ghosts, lists; collective), wall time incl. its host syncs ----
    sync_all()
    t0 = time.perf_counter()
    dom.reneighbor()
    sync_all()
    reneighbor_ms = (time.perf_counter() - t0) * 1e3

    dom.compute(1, 1)
    th1 = dom.thermo()

    value = s.n * job["steps"] / elapsed / 1e6
    ms_per_step = elapsed / job["steps"] * 1e3
    if wl == "rebomos":
        lj = "rebo_lj_gather_kernel" if os.environ.get("MDP_LJ_TILE", "1") == "0" else "rebo_lj_tile_kernel"
        phases = {"rebo_centre3_kernel + rebo_centre_kernel<8|12|16|32> (all launches)": kms[0],
                  "rebo_centre_general_kernel (centres that outgrew their lane group)": kms[1],
                  "tile_prune_kernel (when due)": kms[2], lj + " (one launch)": kms[3]}
        single = {lj: kms[3]}                       # phases that are ONE launch: candidates for `dominant_kernel`
    else:
        dens = "aeam_tile_density_kernel" if os.environ.get("MDP_AEAM_PERSIST", "") == "0" else "aeam_ptile_kernel"
        # (each phase is timed between its own pair of events: no exchange and no host gap lies inside one)
        phases = {dens + " (one launch)": kms[0], "aeam_density_ang_kernel": kms[1], "aeam_embed_kernel": kms[2],
                  "aeam_tile_force_kernel (one launch)": kms[3], "aeam_force_ang_kernel": kms[4]}
        single = {dens: kms[0], "aeam_density_ang_kernel": kms[1], "aeam_embed_kernel": kms[2],
                  "aeam_tile_force_kernel": kms[3], "aeam_force_ang_kernel": kms[4]}
        if dist is not None:   # several GPUs: the tiles that reach no remote ghost run behind the exchanges
            phases[dens + ", interior tiles (position exchange in flight)"] = kms[5]
            phases["aeam_tile_force_kernel, interior tiles (fp / ghost-force exchange in flight)"] = kms[6]
            single[dens + " (interior tiles)"] = kms[5]
            single["aeam_tile_force_kernel (interior tiles)"] = kms[6]
    # algorithmic bytes of ONE pass of the path over this rank's atoms: SURVEY 8(d) per-atom figure x atoms.
    # `achieved` / `frac` use the time of ALL kernels of the path (the contract figure is per atom-step of the
    # whole compute(), not of its longest kernel); the largest single launch is reported beside it with ITS counter bytes.
    alg_bytes = B_ALG[wl] * dom.nlocal
    kall = float(sum(phases.values()))
    path_achieved = alg_bytes / (kall * 1e-3) / 1e9 if kall > 0 else 0.0
    step_achieved = B_ALG[wl] * s.n / world / (ms_per_step * 1e-3) / 1e9
    flops_path = FLOP_ALG[wl] * dom.nlocal / (kall * 1e-3) / 1e12 if kall > 0 else 0.0
    flops_step = FLOP_ALG[wl] * s.n / world / (ms_per_step * 1e-3) / 1e12
    # (counter traffic is stored per lattice configuration: an off-lattice variant of the same size has no entry of its own)
    ent, traffic_note = pmc_entry(wl + ("_" + "_".join(sorted(dis)) if dis else ""), rep, world)
    traffic = ent.get("bytes_per_step") if ent else None
    dname = max(single, key=lambda k: single[k])
    dms = float(single[dname])
    dom_k = {"name": dname, "ms": round(dms, 4), "launches_per_step": 1}
    if ent:
        mine = [(k, v) for k, v in ent.get("per_kernel", {}).items() if k.startswith(dname)]
        if mine and dms > 0:
            kb = max(v["hbm_bytes"] for _, v in mine)     # (the force-only variant is the one measured per step)
            dom_k.update(counter_name=max(mine, key=lambda kv: kv[1]["hbm_bytes"])[0], counter_bytes=kb,
                         counter_GBps=round(kb / (dms * 1e-3) / 1e9, 1),
                         counter_frac=round(kb / (dms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4))
    out = {
        "metric": "Matom-steps/sec",
        "value": round(value, 4),
        "unit": "Matom-steps/s",
        "n_gpus": world,
        "steps": job["steps"],
        "warmup": job["warmup"],
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "ns_per_day": round(job["steps"] / elapsed * 0.001 * 86.4, 4),
        "config": {"workload": wname, "atoms": s.n, "style": wl, "parallelism": f"spatial-dd{world}",
                   "transport": (("library (csrc/comm_rccl.hip, ncclSend/ncclRecv) on the TEST DOUBLE of tests/native: "
                                  f"{world} ranks share one GPU, host-staged -- a rehearsal of the schedule, not a measurement"
                                  if par.get("double") else
                                  ("rccl (library, ncclSend/ncclRecv)" if native else "rccl (torch.distributed all_to_all)")
                                  if not stage_host else par["backend"] + "-staged (rehearsal)")
                                 + (" to the rank itself (one-rank rehearsal)" if par.get("self_remote") else ""))
                   if dist is not None else "none (one GPU)",
                   "overlap_policy": ({k: overlap[k] for k in ("overlap_policy", "overlap_policy_fixed_by_env",
                                                               "overlap_policy_trial_ms", "trial_steps")} if overlap else None),
                   "rccl_library": par.get("rccl_lib") if native else None,
                   "rccl_library_is_test_double": bool(par.get("double")),
                   "initial_temp_K": job["temp"], "skin": skin, "thermo_every": thermo_every,
                   "displacement_check": "every step, deferred on-device flag" if dist is None
                   else ("every step, every rank's flag gathered behind the position exchange (no blocking collective)"
                         if native else f"every {check_every} steps, collective"),
                   "reneighborings_in_timed_region": rebuilds, "reneighbor_wall_ms": round(reneighbor_ms, 3),
                   "dangerous_builds": int(dom.dangerous),
                   "inner_skin": (float(os.environ["MDP_INNER_SKIN"]) if "MDP_INNER_SKIN" in os.environ
                                  else "adaptive from 1.0") if wl == "rebomos" else None,
                   "inner_skin_in_effect": round(lstate["skin"], 3) if wl == "rebomos" else None,
                   "inner_skin_cap": (round(lstate["inner_skin_cap"], 3) or None) if wl == "rebomos" else None,
                   "late_style_list_builds_rank0": lstate["late_builds"],
                   "style_list_builds_in_timed_region_rank0": int(style_builds),
                   "row_prunings_in_timed_region_rank0": prune1["prunings"] - prune0["prunings"],
                   "row_prunings_late_rank0": prune1["late"] - prune0["late"],
                   "row_pruning": (f"tile rows re-filtered to window + {prune1['buffer']:.2f} A from the current positions, "
                                   "own displacement trigger") if prune1["active"] else "off",
                   "pe_per_atom_start_eV": round(pe0 / s.n, 6), "pe_per_atom_end_eV": round(th1["pe"] / s.n, 6),
                   "temp_end_K": round(th1["temp"], 2)},
        "roofline": {"bound": "hbm", "achieved": round(path_achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(path_achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_note": traffic_note,
                     "basis": "algorithmic bytes of one compute() pass of rank 0 / time of ALL kernels of the path "
                              f"(force-only steps, HIP events on the compute stream, mean of {kcount[0]} steps)",
                     "algorithmic_bytes_per_pass": alg_bytes, "path_ms": round(kall, 4),
                     "phase_ms": {n: round(float(m), 4) for n, m in phases.items()},
                     "dominant_kernel": dom_k,
                     "binding": binding_block(wl + ("_" + "_".join(sorted(dis)) if dis else ""), rep, world,
                                              {"rebo_lj_tile_kernel": "lj_tile", "rebo_lj_gather_kernel": "lj_gather",
                                               "aeam_tile_force_kernel": "aeam_tile_force", "aeam_ptile_kernel": "aeam_ptile",
                                               "aeam_tile_density_kernel": "aeam_tile_density"}.get(dname.split(" ")[0], dname)),
                     "whole_step": {"achieved": round(step_achieved, 2), "frac": round(step_achieved / HBM_PEAK_GBPS, 5)},
                     "fp64": {"bound": "fp64-valu", "flop_per_atom_step": FLOP_ALG[wl],
                              "achieved": round(flops_path, 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": round(flops_path / FP64_PEAK_TFLOPS, 5),
                              "whole_step_frac": round(flops_step / FP64_PEAK_TFLOPS, 5)}},
    }
    # how the work was spread over the kernel classes (the lattice-tuned fast paths against their fallbacks)
    try:
        cs = ctx.md_class_stats()
    except AttributeError:       # (an older build of the library, loaded through MDP_LIB_PATH for an A/B run)
        cs = [0] * 32
    if wl == "rebomos":
        names = ("1 lane/centre", "8 lanes", "12 lanes", "16 lanes", "32 lanes")
        out["config"]["work_split"] = {
            "centres_per_class_at_last_list_build": {f"{names[g]} {'Mo' if e == 0 else 'S'}": cs[2 * g + e] + cs[10 + 2 * g + e]
                                                     for g in range(5) for e in range(2) if cs[2 * g + e] + cs[10 + 2 * g + e]},
            "centres_with_a_4th_neighbour_last_step": sum(cs[25:29]),
            "centres_handed_to_the_general_kernel_last_step": cs[24],
            "lj_tiles_small_union_class": cs[20], "lj_tiles_large_union_class": cs[21],
            "largest_union": cs[30], "small_class_union_limit": cs[29]}
    else:
        out["config"]["work_split"] = {"angular_centres_rank0": cs[0], "tiles": cs[1], "largest_union": cs[30],
                                       "density_kernel": dens}
    if dist is not None and wl == "aeam":
        st = ctx.md_aeam_state()
        out["config"].update(aeam_tiles_rank0=st["tiles"], aeam_interior_tiles_rank0=st["interior_tiles"],
                             aeam_ghost_force_exchange=bool(getattr(dom, "ghost_forces", False)),
                             aeam_steps_with_exchanges_behind_interior_tiles_rank0=int(dom.aeam_overlapped),
                             aeam_step="position exchange behind the density of the interior tiles; fp forward and "
                                       "ghost-force reverse exchange behind the pair forces of the interior tiles")
    if dist is not None:
        # per-rank shape of the decomposition (log.rebomos-bulk.4:72-75 prints the same per-rank counts)
        mine = torch.tensor([dom.nlocal, dom.nself, dom.nrecv, dom.nsend], dtype=torch.int64, device="cpu" if stage_host else dev)
        rows = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(rows, mine)
        m = torch.stack(rows).cpu().numpy()
        out["config"].update(nlocal_per_rank=m[:, 0].tolist(), self_ghosts_per_rank=m[:, 1].tolist(),
                             remote_ghosts_per_rank=m[:, 2].tolist(), halo_bytes_sent_per_step_per_rank=(m[:, 3] * 24).tolist(),
                             procgrid=list(dom.grid))
    ctx.close()
    # (keep: the AEAM tables point into the potential-file object `af`, so it must outlive every user of `pot`)
    return out, dict(s=s, pot=pot, skin=skin, cutghost=cutghost, keep=getattr(ctx, "_af", None))


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
    ap.add_argument("--strain", type=float, nargs=2, default=None, metavar=("SCALE", "JITTER"),
                    help="rebomos: box and coordinates x SCALE, uniform jitter +-JITTER A (R-strain-112: 1.12 0.15)")
    ap.add_argument("--frac2", type=float, default=None, help="aeam: fraction of atoms of type 2 (sample.in: 0.0075)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-mode", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary block (REBO-MoS from 300 K, AEAM configuration #3) of the default one-GPU run")
    ap.add_argument("--inner-skin", type=float, default=None,
                    help="skin of the style's own device-built lists in A (default: library default 1.0, capped by "
                         "the host skin); they are rebuilt on the device when an atom has moved half of it")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))      # the parent has made no GPU call; the children are fresh processes
    # stdout carries ONE line, the result.  Libraries print there too (RCCL writes a version banner to stdout when a
    # communicator comes up), so file descriptor 1 points at stderr until the line is printed.
    sys.stdout.flush()
    fd_out = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: refusing to measure another rank count")
    custom = (args.replicate is not None or args.workload != "rebomos" or args.temp != 0.0 or args.strain is not None
              or args.frac2 is not None)
    if args.replicate is None:
        args.replicate = DEFAULT_REPLICATE[args.workload]
    if args.inner_skin is not None:
        os.environ["MDP_INNER_SKIN"] = str(args.inner_skin)

    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    graft.load_package()
    from lammps_plugins_amd.host import capi, resident, system as S
    E = dict(capi=capi, resident=resident, S=S)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # MDP_BENCH_BACKEND=gloo is a rehearsal switch for boxes with fewer GPUs than ranks: every rank uses
    # GPU (local_rank mod #GPUs) and the halo is staged through the host.  The judged runs use RCCL.
    # MDP_RCCL_LIBRARY pointing at the TEST DOUBLE of tests/native (several ranks on one GPU through the library's own
    # transport, host-staged): the process group is gloo then -- it only hands the communicator id round and reduces
    # the clock -- and the line is marked as a rehearsal.
    rccl_lib, double = (capi.comm_library() if os.environ.get("MDP_RCCL_LIBRARY") else ("librccl.so.1", False))
    backend = os.environ.get("MDP_BENCH_BACKEND", "gloo" if double else "nccl")
    stage_host = backend != "nccl"
    ngpu = torch.cuda.device_count()
    if stage_host:
        local_rank = local_rank % ngpu
    elif ngpu < world:
        raise SystemExit(f"bench.py: {world} ranks asked for, this box has {ngpu} GPU(s): refusing to share a GPU between "
                         "RCCL ranks (MDP_BENCH_BACKEND=gloo is the one-GPU rehearsal)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist, rccl_ranks = None, 1
    # MDP_BENCH_SELF_REMOTE=1 (rehearsal on one GPU): a ONE-rank run takes the N>1 code path -- process group on RCCL,
    # collective displacement checks, per-rank gathers -- and every periodic self-image travels through the all-to-all
    # to the rank itself.  The line says so (config.transport); it is not the one-GPU headline.
    self_remote = world == 1 and os.environ.get("MDP_BENCH_SELF_REMOTE", "0") not in ("", "0")
    if world > 1 or self_remote:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if stage_host:
            dist.init_process_group(backend, rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        one = torch.ones(1, dtype=torch.float64, device="cpu" if stage_host else dev)
        dist.all_reduce(one)                   # first collective: how many ranks does the backend really connect?
        rccl_ranks = int(round(float(one.item())))
        if rccl_ranks != world or dist.get_world_size() != world:
            raise SystemExit(f"bench.py: {world} ranks started, the collective saw {rccl_ranks}: no result")
    # transport of the bricks' exchanges: "native" (the default) = grouped ncclSend/ncclRecv inside libmdpair_hip.so
    # (csrc/comm_rccl.hip: whole steps in two library calls, the `check yes` decision riding with the halo; the process
    # group then only distributes the communicator id); MDP_BENCH_TRANSPORT=torch = all-to-all through torch.distributed
    # (RCCL) with a collective displacement check every --check-every steps
    native = dist is not None and os.environ.get("MDP_BENCH_TRANSPORT", "native") == "native" and (not stage_host or double)
    if native and os.environ.get("MDP_BENCH_TEST_FAIL_NATIVE", "0") not in ("", "0"):
        raise SystemExit("bench.py: MDP_BENCH_TEST_FAIL_NATIVE set -- the library-transport ranks give up (test of the fallback)")
    par = dict(world=world, rank=rank, local_rank=local_rank, dist=dist, dev=dev, stage_host=stage_host, native=native,
               backend=backend, self_remote=self_remote, double=double and native, rccl_lib=rccl_lib)

    job = dict(workload=args.workload, replicate=list(args.replicate), temp=args.temp, steps=args.steps, warmup=args.warmup,
               thermo_every=THERMO_EVERY[args.workload] if args.thermo is None else args.thermo, check_every=args.check_every)
    if args.strain is not None and args.workload == "rebomos":
        job["disorder"] = dict(scale=args.strain[0], jitter=args.strain[1], seed=1234)
    if args.frac2 is not None and args.workload == "aeam":
        job["disorder"] = dict(frac2=args.frac2)
    out, pieces = run_job(E, job, par)
    out["config"]["rccl_ranks"] = rccl_ranks
    if os.environ.get("MDP_BENCH_FALLBACK_REASON"):
        out["transport_fallback"] = os.environ["MDP_BENCH_FALLBACK_REASON"]
    if rank == 0 and world == 1 and dist is None and not args.no_host_mode:
        try:
            hm, img = host_mode_rate(E, pieces["s"], args.workload, pieces["pot"], pieces["skin"], pieces["cutghost"])
            out["host_mode_ms_per_step"] = round(hm, 3)
            out["host_mode_images_on_device"] = img
            out["host_mode_Matom_steps_per_s"] = round(pieces["s"].n / hm / 1e3, 2)
        except Exception as e:  # noqa: BLE001 -- informational figure
            log(f"[bench] host-mode measurement failed: {e}")
    pieces = None

    # ---- secondary block of the default one-GPU run: what the cold-start headline does not show ----
    if world == 1 and dist is None and not custom and not args.no_secondary:
        sec = {}
        try:
            hot = dict(job, temp=300.0, steps=max(200, args.steps), warmup=20)
            o, _ = run_job(E, hot, par)
            sec["rebomos_300K"] = {k: o[k] for k in ("value", "unit", "steps", "warmup", "ms_per_step", "config", "roofline")}
            sec["rebomos_300K"]["note"] = ("the headline system started at 300 K (SURVEY 8d config #4 variant): style-list "
                                           "builds and row prunings happen inside the timed region")
        except Exception as e:  # noqa: BLE001
            log(f"[bench] secondary REBO-MoS 300 K run failed: {e}")
        try:
            aj = dict(workload="aeam", replicate=DEFAULT_REPLICATE["aeam"], temp=863.0, steps=1000, warmup=20,
                      thermo_every=THERMO_EVERY["aeam"], check_every=args.check_every)
            o, pieces = run_job(E, aj, par)
            o["note"] = "BASELINE.json configs[2] / SURVEY 8d config #3 as written: 1,000,188 atoms, 863 K, 1000 NVE steps, check every step"
            if not args.no_host_mode:
                try:
                    hm, img = host_mode_rate(E, pieces["s"], "aeam", pieces["pot"], pieces["skin"], pieces["cutghost"])
                    o["host_mode_ms_per_step"] = round(hm, 3)
                    o["host_mode_images_on_device"] = img
                    o["host_mode_Matom_steps_per_s"] = round(pieces["s"].n / hm / 1e3, 2)
                except Exception as e:  # noqa: BLE001
                    log(f"[bench] aeam host-mode measurement failed: {e}")
            pieces = None
            if not args.no_cpu_baseline:
                attach_cpu_baseline(o, "aeam", 5.0)
            for k in ("metric", "higher_is_better", "scaling", "vs_baseline", "data"):
                o.pop(k, None)
            sec["aeam_config3"] = o
        except Exception as e:  # noqa: BLE001
            log(f"[bench] secondary AEAM run failed: {e}")
        # ---- speed OFF the perfect lattice (the kernels' lane-group classes, one-lane S centres and element-sorted
        # tiles are tuned to 2H-MoS2 / fcc): the strained, jittered cell of SURVEY Appendix C at full size, and the alloy
        # with ten times the angular atoms.  Same JSON shape; `config.work_split` says which kernels carried the load.
        try:
            sj = dict(job, temp=300.0, steps=max(200, args.steps), warmup=20,
                      disorder=dict(scale=1.12, jitter=0.15, seed=1234))
            o, _ = run_job(E, sj, par)
            sec["rebomos_strained_112_jitter_300K"] = {k: o[k] for k in ("value", "unit", "steps", "warmup", "ms_per_step", "config", "roofline")}
            sec["rebomos_strained_112_jitter_300K"]["note"] = (
                "R-strain-112 of SURVEY Appendix C replicated 24x24x24 (3 981 312 atoms), 300 K: switching interior, LJ cubic "
                "branch, coordination spread -- overflow lists, general kernel and second tile launch class carry load")
        except Exception as e:  # noqa: BLE001
            log(f"[bench] secondary strained REBO-MoS run failed: {e}")
        try:
            aj = dict(workload="aeam", replicate=DEFAULT_REPLICATE["aeam"], temp=863.0, steps=300, warmup=20,
                      thermo_every=THERMO_EVERY["aeam"], check_every=args.check_every, disorder=dict(frac2=0.08))
            o, _ = run_job(E, aj, par)
            sec["aeam_8pct_si"] = {k: o[k] for k in ("value", "unit", "steps", "warmup", "ms_per_step", "config", "roofline")}
            sec["aeam_8pct_si"]["note"] = ("A-6-8pct of SURVEY Appendix C at the size of config #3: 1 000 188 atoms, 8 % Si "
                                           "(80 000 angular centres with their O(n^2) triplet loops), 863 K")
        except Exception as e:  # noqa: BLE001
            log(f"[bench] secondary AEAM 8 % Si run failed: {e}")
        # ---- the drop-in path itself, in the driver's record: `plugin load` + `fix nve/mdp` on the mini-host (a fresh
        # child process; every context of this process is closed), inputs with the REFERENCE's neighbor settings
        # (in.rebomos-bulk:27-34: no neigh_modify line; sample.in:17-18)
        if not args.no_host_mode:
            try:
                pl = plugin_load_run("in.rebomos-4m.nve-mdp.mi355x", {"thermo 50": "thermo 100", "run 100": "run 300"})
                ref = resident_thermo_rows(E, (0, 100))
                got = {int(r[0]): r[1:5] for r in pl["thermo_rows"]}
                worst = 0.0
                for r in ref:
                    g = got.get(int(r[0]))
                    if g is None:
                        worst = float("inf")
                        continue
                    for a, b in zip(g, r[1:5]):
                        worst = max(worst, abs(a - b) / max(abs(b), 1.0))
                pl["thermo_rows_vs_resident_run"] = {"steps": [int(r[0]) for r in ref], "columns": "temp press pe ke",
                                                     "max_rel_diff": float(worst), "equal_to_printed_digits": bool(worst < 5e-8)}
                pl["note"] = ("BASELINE.json configs[3] on one GPU through the reference's plugin surface: minilmp, plugin load "
                              "rebomosplugin.so, pair_style rebomos, fix nve/mdp, the reference's neighbor settings unchanged")
                sec["plugin_load_nve_mdp"] = pl
            except Exception as e:  # noqa: BLE001
                log(f"[bench] plugin-load REBO-MoS run failed: {e}")
            try:
                pl = plugin_load_run("in.aeam-alsi.nve-mdp.mi355x", {})
                pl["note"] = ("USER-AEAM/sample.in's system (32 000 atoms, 0.75 % Si by the mini-host's own RNG, NVE from 863 K) through "
                              "plugin load aeamplugin.so + fix nve/mdp, neigh_modify every 1 delay 1 check yes as sample.in:18")
                sec["plugin_load_nve_mdp_aeam"] = pl
            except Exception as e:  # noqa: BLE001
                log(f"[bench] plugin-load AEAM run failed: {e}")
            try:   # `bricks yes`: the same inputs with the library's decomposition (one brick) running the steps -- lists and
                   # reneighborings on the device too, the host's Neighbor idle for the length of the run
                plb = plugin_load_run("in.rebomos-4m.nve-mdp.mi355x", {"thermo 50": "thermo 100", "run 100": "run 300",
                                                                        "fix integrate all nve/mdp": "fix integrate all nve/mdp bricks yes"})
                base_rows = {int(r[0]): r[1:5] for r in sec.get("plugin_load_nve_mdp", {}).get("thermo_rows", [])}
                worst = max([abs(a - b) / max(abs(b), 1.0) for r in plb["thermo_rows"] if int(r[0]) in base_rows
                             for a, b in zip(r[1:5], base_rows[int(r[0])])] or [float("inf")])
                plb["thermo_rows_vs_default_fix"] = {"max_rel_diff": float(worst), "equal_to_printed_digits": bool(worst < 5e-8)}
                plb["note"] = "as plugin_load_nve_mdp with `fix integrate all nve/mdp bricks yes`"
                sec["plugin_load_nve_mdp_bricks"] = plb
                pla = plugin_load_run("in.aeam-alsi.nve-mdp.mi355x", {"region MeSi block 0 20 0 20 0 20": "region MeSi block 0 63 0 63 0 63",
                                                                       "run 400": "run 300",
                                                                       "fix integrate all nve/mdp": "fix integrate all nve/mdp bricks yes"})
                pla["note"] = ("BASELINE.json configs[2]'s size (63^3 fcc cells, 1 000 188 atoms, 0.75 % Si, NVE from 863 K, neigh_modify every 1 "
                               "delay 1 check yes) through plugin load aeamplugin.so + fix nve/mdp bricks yes: every reneighboring on the device")
                sec["plugin_load_nve_mdp_aeam1m_bricks"] = pla
            except Exception as e:  # noqa: BLE001
                log(f"[bench] plugin-load runs with bricks yes failed: {e}")
            try:   # the same system driven by the C++ host of minihost/ddhost.cpp: resident mode through the C-ABI, no Python in the loop
                pkg = os.path.join(ROOT, "lammps-plugins_amd")
                p = subprocess.run([os.path.join(pkg, "ddhost"), "-ranks", "1", "-replicate", *map(str, DEFAULT_REPLICATE["rebomos"]),
                                    "-steps", "200", "-thermo", str(THERMO_EVERY["rebomos"])], cwd=pkg, capture_output=True, text=True, timeout=600)
                import re
                m = re.search(r"Loop time of ([0-9.eE+-]+) on 1 procs for (\d+) steps with (\d+) atoms", p.stdout)
                if p.returncode != 0 or not m:
                    raise RuntimeError(f"ddhost failed ({p.returncode}): {p.stderr[-300:]}")
                rows = [[float(w) for w in l.split()] for l in p.stdout.splitlines() if re.match(r"^\s*\d+\s+[-0-9.e+]+\s+[-0-9.e+]+", l)]
                sec["cpp_host_resident"] = dict(
                    ms_per_step=round(float(m.group(1)) / int(m.group(2)) * 1e3, 4), steps=int(m.group(2)), atoms=int(m.group(3)),
                    Matom_steps_per_s=round(int(m.group(3)) * int(m.group(2)) / float(m.group(1)) / 1e6, 2),
                    thermo_every=THERMO_EVERY["rebomos"], pe_step0=rows[0][3] if rows else None,
                    pe_step0_over_13824_cells=(rows[0][3] / 13824.0 if rows else None),
                    note="minihost/ddhost.cpp: the headline system, device-resident NVE with thermo every 10 steps as in.rebomos-bulk:31, "
                         "driven by a C++ program through include/mdpair_hip.h alone (mdp_md_integrate_check / mdp_md_compute per step); "
                         "PE of step 0 / 13 824 cells = log.rebomos-bulk.1:54")
            except Exception as e:  # noqa: BLE001
                log(f"[bench] ddhost run failed: {e}")
        out["secondary"] = sec
    # ---- N ranks: the two C++ hosts on the same N GPUs, as children of rank 0 after the timed region (the other ranks wait
    # at the barrier below).  One thread per GPU in ONE process each: `ddhost -ranks N` through the C-ABI alone, and
    # `minilmp -np N` through the plugin surface with fix nve/mdp on the library's bricks.  Informational; a failure is logged.
    plain_rebomos = args.workload == "rebomos" and args.temp == 0.0 and args.strain is None   # (any --replicate: the children take it)
    if world > 1 and rank == 0 and plain_rebomos and not args.no_secondary and native:
        sec = {}
        pkg = os.path.join(ROOT, "lammps-plugins_amd")
        import re
        try:
            p = subprocess.run([os.path.join(pkg, "ddhost"), "-ranks", str(world), "-replicate", *map(str, args.replicate),
                                "-steps", "100", "-thermo", str(THERMO_EVERY["rebomos"])], cwd=pkg, capture_output=True, text=True, timeout=300)
            m = re.search(r"Loop time of ([0-9.eE+-]+) on (\d+) procs for (\d+) steps with (\d+) atoms", p.stdout)
            if p.returncode != 0 or not m:
                raise RuntimeError(f"ddhost failed ({p.returncode}): {p.stderr[-300:]}")
            pol = re.search(r"Overlap policy = (\w+)", p.stdout)
            sec["cpp_host_resident_ranks"] = dict(
                ranks=int(m.group(2)), ms_per_step=round(float(m.group(1)) / int(m.group(3)) * 1e3, 4), steps=int(m.group(3)),
                atoms=int(m.group(4)), Matom_steps_per_s=round(int(m.group(4)) * int(m.group(3)) / float(m.group(1)) / 1e6, 2),
                overlap_policy=pol.group(1) if pol else None, test_double="TEST DOUBLE" in p.stdout,
                note="minihost/ddhost.cpp: the same system on the same GPUs, one host thread per GPU in one process, through "
                     "include/mdpair_hip.h alone; the first steps are the overlap-policy trial (inside the loop time)")
        except Exception as e:  # noqa: BLE001
            log(f"[bench] ddhost on {world} ranks failed: {e}")
        try:
            rep = " ".join(map(str, args.replicate))
            pl = plugin_load_run("in.rebomos-4m.nve-mdp.mi355x", {"replicate 24 24 24": "replicate " + rep, "thermo 50": "thermo 100",
                                                                    "run 100": "run 200"}, timeout=420, np=world)
            pl["note"] = ("the same system through the reference's plugin surface on N ranks of the mini-host (threads, one GPU each): "
                          "plugin load, pair_style rebomos, fix nve/mdp on the library's bricks (INTEGRATION.md section 2)")
            sec["plugin_load_nve_mdp_ranks"] = pl
        except Exception as e:  # noqa: BLE001
            log(f"[bench] plugin-load run on {world} ranks failed: {e}")
        if sec:
            out["secondary"] = sec
    # the CPU baseline runs on rank 0 AFTER every timed region (the other ranks wait at the barrier below)
    if rank == 0 and not args.no_cpu_baseline:
        attach_cpu_baseline(out, args.workload, 8.0)
    if rank == 0:
        sys.stdout.flush()
        os.dup2(fd_out, 1)
        print(json.dumps(out, default=_jsonable), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
