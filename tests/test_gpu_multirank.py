"""GPU, N>1 path: two processes share cuda:0, each owns one brick of the box (RankDomain), ghost
positions travel through pack_x_kernel -> all_to_all -> unpack_x_kernel every step.  The transport
here is gloo staged through the host (one GPU cannot host two RCCL ranks); everything else -- plan,
device pack/unpack, owner-computes forces, AEAM's fp forward and force reverse exchanges, thermo
reduction -- is the code bench.py runs with RCCL.  Checked against the single-domain GPU run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import POT_AEAM, POT_REBOMOS
import hostplan

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(style):
    from lammps_plugins_amd.host import capi, system as S
    if style == "rebomos":
        s = S.jitter(S.replicate(S.rebomos_bulk_cell(), (2, 1, 1)), 0.05, seed=3)
        v0 = S.gaussian_velocities(s, 300.0, seed=5)
        return s, v0
    s = S.jitter(S.fcc_cell(4.045, (8, 6, 6), frac_type2=0.03, seed=12), 0.03, seed=13)
    af = capi.AeamFile(POT_AEAM)
    s.mass[1:3] = af.mass
    return s, S.gaussian_velocities(s, 500.0, seed=7)


def _setup_ctx(style):
    from lammps_plugins_amd.host import capi
    ctx = capi.Context(0)
    keep = None
    if style == "rebomos":
        p = capi.read_rebomos_file(POT_REBOMOS)
        ctx.rebomos_set_params(p)
        return ctx, capi.STYLE_REBOMOS, 3.0 * p.rcmax[0][0] + 2.0, 2.0, [0, 0, 1], keep
    af = capi.AeamFile(POT_AEAM)
    tabs = af.build()
    ctx.aeam_set_tables(tabs)
    return ctx, capi.STYLE_AEAM, float(af.cut_table(tabs).max()) + 1.0, 1.0, None, (af, tabs)


def _run(dom, nsteps, dist_mod=None):
    dom.build_neighbors()
    dom.compute(eflag=1, vflag=1)
    rows = [dom.ctx.md_thermo()]
    for k in range(nsteps):
        dom.step(0, 0)
    dom.compute(eflag=1, vflag=1)
    rows.append(dom.ctx.md_thermo())
    return rows


def _worker_device(rank, world, port, style, q):
    """the path bench.py runs: DeviceDomain + resident.Transport (here on a gloo group, staged through the host),
    with a forced reneighboring in the middle so that atoms migrate through the torch.distributed all-to-all"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import conftest  # noqa: F401
    from lammps_plugins_amd.host import resident
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        s, v0 = _build(style)
        v0 = v0 + np.array([150.0, 40.0, -60.0])          # a drift that carries atoms across the brick faces
        ctx, st, cutghost, skin, map_, keep = _setup_ctx(style)
        tr = resident.Transport(dist, torch.device("cuda", 0), stage_host=True)
        dom = resident.DeviceDomain(ctx, st, s, cutghost, skin, map_, v0=v0, transport=tr)
        dom.compute(1, 1)
        rows = [dom.thermo()]
        left = 0
        for k in range(1, 25):
            rb = k % 4 == 0
            dom.step(0, 0, rebuild=rb)
            if rb:
                left += ctx.dd_info()["left_last"]
        dom.compute(1, 1)
        rows.append(dom.thermo())
        got = ctx.md_download(dom.nlocal, want=("x", "f"))
        q.put((rank, dom.tags_local, got["x"], got["f"], [[r["ke"], r["pe"], *r["virial"]] for r in rows], dom.nlocal, left))
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("style", ["rebomos", "aeam"])
def test_two_ranks_device_domain_over_torch_distributed(style):
    from lammps_plugins_amd.host import resident, system as S
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_worker_device, args=(r, 2, port, style, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s, v0 = _build(style)
    v0 = v0 + np.array([150.0, 40.0, -60.0])
    ctx, st, cutghost, skin, map_, keep = _setup_ctx(style)
    dom = resident.DeviceDomain(ctx, st, s, cutghost, skin, map_, v0=v0)
    dom.compute(1, 1)
    rows = [dom.thermo()]
    for k in range(1, 25):
        dom.step(0, 0, rebuild=k % 4 == 0)
    dom.compute(1, 1)
    rows.append(dom.thermo())
    got = ctx.md_download(dom.nlocal, want=("x", "f"))
    tags = dom.tags_local
    x1 = np.zeros((s.n, 3)); f1 = np.zeros((s.n, 3))
    x1[tags - 1] = got["x"]; f1[tags - 1] = got["f"]
    x2 = np.zeros((s.n, 3)); f2 = np.zeros((s.n, 3))
    assert sum(r[5] for r in res) == s.n
    assert sum(r[6] for r in res) > 10                    # atoms changed owner through the all-to-all
    for rank, tg, x, f, tot, nloc, left in res:
        x2[tg - 1] = x
        f2[tg - 1] = f
    tot = res[0][4]
    for k in range(2):
        assert tot[k][1] == pytest.approx(rows[k]["pe"], rel=1e-11)
        assert tot[k][0] == pytest.approx(rows[k]["ke"], rel=1e-9, abs=1e-12)
        assert np.allclose(tot[k][2:], rows[k]["virial"], rtol=1e-8, atol=1e-6)
    dx = x2 - x1
    dx -= np.round(s.box.x2lamda(dx + s.box.lo)) @ s.box.h.T
    assert np.abs(dx).max() < 1e-9
    assert np.abs(f2 - f1).max() < 1e-7
    ctx.close()


def _worker(rank, world, port, style, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import conftest  # noqa: F401
    from lammps_plugins_amd.host import resident
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        s, v0 = _build(style)
        ctx, st, cutghost, skin, map_, keep = _setup_ctx(style)
        dom = hostplan.make_domain(ctx, st, s, cutghost, skin, map_, v0=v0, dist=dist,
                                   device=torch.device("cuda", 0), stage_host=True)
        rows = _run(dom, 25)
        tot = []
        for r in rows:
            t = torch.tensor([r["ke"], r["pe"]] + list(r["virial"]), dtype=torch.float64)
            dist.all_reduce(t)
            tot.append(t.numpy())
        got = ctx.md_download(dom.nlocal, want=("x", "f"))
        q.put((rank, dom.tags_local, got["x"], got["f"], tot, dom.nlocal, dom.nghost))
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("style", ["rebomos", "aeam"])
def test_two_ranks_on_one_gpu_match_single_domain(style):
    from lammps_plugins_amd.host import resident
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_worker, args=(r, 2, port, style, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-domain reference on the same GPU
    s, v0 = _build(style)
    ctx, st, cutghost, skin, map_, keep = _setup_ctx(style)
    dom = hostplan.make_domain(ctx, st, s, cutghost, skin, map_, v0=v0)
    rows = _run(dom, 25)
    got = ctx.md_download(dom.nlocal, want=("x", "f"))
    x1 = np.zeros((s.n, 3)); f1 = np.zeros((s.n, 3))
    x1[dom.tags_local - 1] = got["x"]; f1[dom.tags_local - 1] = got["f"]
    x2 = np.zeros((s.n, 3)); f2 = np.zeros((s.n, 3))
    assert sum(r[5] for r in res) == s.n
    for rank, tags, x, f, tot, nloc, ngh in res:
        x2[tags - 1] = x
        f2[tags - 1] = f
    tot = res[0][4]
    for k in range(2):
        assert tot[k][1] == pytest.approx(rows[k]["pe"], rel=1e-11)            # PE
        assert tot[k][0] == pytest.approx(rows[k]["ke"], rel=1e-9, abs=1e-12)  # KE
        assert np.allclose(tot[k][2:], rows[k]["virial"], rtol=1e-8, atol=1e-6)
    assert np.abs(x2 - x1).max() < 1e-9     # 25 steps of identical dynamics
    assert np.abs(f2 - f1).max() < 1e-7
    ctx.close()


def _worker_rccl_self(port, style, q):
    """the torch transport of bench.py (MDP_BENCH_TRANSPORT=torch; the default since round 5 is the library's own,
    tests/test_gpu_domain.py) -- resident.Transport on the "nccl" backend (RCCL), device buffers straight into
    all_to_all_single, the exchange asynchronous behind the interior centres -- with a one-rank process group:
    `self_remote` makes every periodic self-image a remote ghost that travels through the all-to-all to the rank itself"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import conftest  # noqa: F401
    from lammps_plugins_amd.host import resident
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        s, v0 = _build(style)
        v0 = v0 + np.array([150.0, 40.0, -60.0])
        out = []
        for remote in (False, True):
            ctx, st, cutghost, skin, map_, keep = _setup_ctx(style)
            tr = resident.Transport(dist, dev, stage_host=False) if remote else None
            dom = resident.DeviceDomain(ctx, st, s, cutghost, skin, map_, v0=v0, transport=tr, self_remote=remote)
            if remote:
                assert dom.nself == 0 and dom.nrecv > 0 and dom.nsend == dom.nrecv
            dom.compute(1, 1)
            rows = [dom.thermo()]
            for k in range(1, 25):
                dom.step(0, 0, rebuild=k % 4 == 0)
            dom.compute(1, 1)
            rows.append(dom.thermo())
            got = ctx.md_download(dom.nlocal, want=("x", "f"))
            order = np.argsort(dom.tags_local)
            out.append((got["x"][order], got["f"][order], [[r["ke"], r["pe"], *r["virial"]] for r in rows], dom.nrecv))
            ctx.close()
        q.put(out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("style", ["rebomos", "aeam"])
def test_default_rccl_transport_with_one_rank(style):
    """what the N-GPU bench line runs per step (pack -> all_to_all_single on RCCL, asynchronous || interior centres ->
    wait -> unpack; counts through all_gather; migration and border records at every reneighboring), on one GPU: the
    trajectory must equal the plain one-GPU run"""
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    p = ctxm.Process(target=_worker_rccl_self, args=(_free_port(), style, q))
    p.start()
    (xp, fp, rp, _), (xr, fr, rr, nrecv) = q.get(timeout=500)
    p.join(timeout=60)
    assert p.exitcode == 0 and nrecv > 0
    s, _ = _build(style)
    dx = xr - xp
    dx -= np.round(s.box.x2lamda(dx + s.box.lo)) @ s.box.h.T
    assert np.abs(dx).max() < 1e-9 and np.abs(fr - fp).max() < 1e-7
    for k in range(2):
        assert rr[k][1] == pytest.approx(rp[k][1], rel=1e-10)
        assert rr[k][0] == pytest.approx(rp[k][0], rel=1e-9, abs=1e-12)
