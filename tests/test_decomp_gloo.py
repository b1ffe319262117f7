"""N>1 path on CPU: world_size-2 gloo run of the domain decomposition + halo plan
(lammps-plugins_amd/host/decomp.py).  Forward exchange of ghost positions, reverse exchange of
ghost forces, decomposition invariance of energy and forces (the reference's own evidence is
log.rebomos-bulk.4:54-56 == log.rebomos-bulk.1:54-56).  Forces come from the CPU oracle here; the
GPU ranks use the same plan with the device pack/unpack kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import POT_REBOMOS
from lammps_plugins_amd.host import decomp, system as S


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _system():
    s = S.replicate(S.rebomos_bulk_cell(), (2, 1, 1))
    return S.jitter(s, 0.08, seed=21)


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import conftest  # noqa: F401  (registers the package)
    import oracle_bindings as ob
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = ob.load()
        P = orc.rebomos_params(POT_REBOMOS)
        s = _system()
        cut = P.cut3rebo + 2.0
        xw = S.wrap(s.box, s.x)
        dec = decomp.Decomposition(s.box, xw, world, cut)
        plan = dec.plan(rank)
        halo = decomp.Halo(plan, "cpu", dist)
        nloc, ngh = len(plan.owned), len(plan.ghost_global)
        # start from the initial ghost positions, then "integrate": move owned atoms
        disp = 0.3 * np.sin(np.arange(s.n)[:, None] * np.array([0.7, 1.3, 2.1]))
        xnew_global = xw + disp
        x_owned = xnew_global[plan.owned]
        xg = np.zeros((ngh, 3))
        # self images refreshed locally (ghost_refresh_kernel)
        selfm = plan.ghost_owner_local >= 0
        xg[selfm] = x_owned[plan.ghost_owner_local[selfm]] + plan.ghost_shift[selfm]
        # forward comm (pack_x_kernel / all_to_all_single / unpack_x_kernel)
        halo.send3[:halo.nsend * 3] = torch.from_numpy((x_owned[plan.send_local] + plan.send_shift).ravel())
        halo.forward3()
        xg[plan.nself:] = halo.recv3[:halo.nrecv * 3].numpy().reshape(-1, 3)
        expect = xnew_global[plan.ghost_global] + plan.ghost_shift
        err_fwd = float(np.abs(xg - expect).max())

        # forces on this rank with reference semantics (scatter incl. ghosts), then reverse comm
        x_all = np.concatenate([x_owned, xg])
        type_all = np.concatenate([s.type[plan.owned], s.type[plan.ghost_global]]).astype(np.int32)
        tag_all = np.concatenate([s.tag[plan.owned], s.tag[plan.ghost_global]]).astype(np.int32)
        rcmax = np.array([[P.rcmax[a][b] for b in range(2)] for a in range(2)])
        cg = np.zeros((3, 3))
        cg[1:, 1:] = rcmax + 2.0
        nn, off, nb = S.neighbor_lists_cpu(x_all, type_all, nloc, cut, cg)
        o = orc.rebomos_compute(P, nloc, x_all, type_all - 1, tag_all, nn, off, nb, eflag=1, vflag=1)
        f = o["f"][:nloc].copy()
        fg = o["f"][nloc:]
        np.add.at(f, plan.ghost_owner_local[selfm], fg[selfm])             # fold_self_ghost_f_kernel
        halo.recv3[:halo.nrecv * 3] = torch.from_numpy(fg[plan.nself:].ravel())   # pack_ghost_f
        halo.reverse3()
        np.add.at(f, plan.send_local, halo.send3[:halo.nsend * 3].numpy().reshape(-1, 3))  # unpack_add_f
        tot = torch.tensor([o["eng"]] + list(o["virial_fdotr"]), dtype=torch.float64)
        dist.all_reduce(tot)
        q.put((rank, plan.owned, f, tot.numpy(), err_fwd, int(nn[:nloc].sum()), nloc, ngh))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_decomposition_matches_single_domain(oracle):
    import mdref
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s = _system()
    P = oracle.rebomos_params(POT_REBOMOS)
    disp = 0.3 * np.sin(np.arange(s.n)[:, None] * np.array([0.7, 1.3, 2.1]))
    s1 = S.System(s.box, S.wrap(s.box, s.x) + disp, s.type, s.tag, s.mass)
    ref = mdref.RebomosCPU(oracle, P, S.System(s.box, S.wrap(s.box, s1.x), s.type, s.tag, s.mass))
    o = ref.compute(ref.s.x)
    f = np.zeros((s.n, 3))
    nn_total = 0
    for rank, owned, fr, tot, err_fwd, nn, nloc, ngh in res:
        assert err_fwd < 1e-12
        f[owned] = fr
        nn_total += nn
        assert tot[0] == pytest.approx(o["eng"], rel=1e-12)
        assert np.allclose(tot[1:], o["virial_fdotr"], rtol=1e-9, atol=1e-7)
    assert sum(r[6] for r in res) == s.n
    assert np.abs(f - o["f_owned"]).max() < 1e-10


def test_plans_are_consistent_for_1_2_4_8_ranks():
    """send/recv counts match pairwise, every atom is owned exactly once, 4-rank counts match
    the reference's 4-rank log for the 288-atom cell (72 atoms/rank, 35712 neighbors/rank)"""
    s = S.rebomos_bulk_cell()
    xw = S.wrap(s.box, s.x)
    for n in (1, 2, 4, 8):
        dec = decomp.Decomposition(s.box, xw, n, 13.4)
        plans = [dec.plan(r) for r in range(n)]
        assert sum(len(p.owned) for p in plans) == s.n
        for a in range(n):
            for b in range(n):
                assert plans[a].send_counts[b] == plans[b].recv_counts[a]
            assert plans[a].recv_counts[a] == 0
        if n == 1:
            assert len(plans[0].ghost_global) == 4285        # log.rebomos-bulk.1:74
        if n == 4:
            assert [len(p.owned) for p in plans] == [72] * 4  # log.rebomos-bulk.4:72
            # log.rebomos-bulk.4:74-75: "Nghost: 2771.5 ave 2775 max 2768 min", histogram 2 | 2
            assert sorted(len(p.ghost_global) for p in plans) == [2768, 2768, 2775, 2775]


# ---- the transport bench.py uses between the bricks (resident.Transport), world_size 2, gloo, CPU tensors -------------
def _transport_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from lammps_plugins_amd.host import resident
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tr = resident.Transport(dist, torch.device("cpu"), stage_host=True)
        # ragged all-to-all: rank r sends (r + 1) * (q + 2) records of width 3 to rank q (nothing to itself)
        sc = np.array([(rank + 1) * (q_ + 2) if q_ != rank else 0 for q_ in range(world)], dtype=np.int64)
        rc = tr.counts(sc)
        send = torch.cat([torch.full((int(sc[q_]) * 3,), 100.0 * rank + q_, dtype=torch.float64) for q_ in range(world)])
        recv, _ = tr.exchange(send, sc, rc, 3)
        flag_any = tr.any(rank == 1)
        tot = tr.sum([1.0 + rank, 10.0])
        q.put((rank, rc.tolist(), recv[:int(rc.sum()) * 3].tolist(), flag_any, tot.tolist()))
    finally:
        dist.destroy_process_group()


def test_transport_ragged_all_to_all_world2():
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_transport_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r[0]: r for r in (q.get(timeout=120) for _ in procs)}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # rank 0 receives what rank 1 addressed to it: (1 + 1) * (0 + 2) = 4 records of value 100
    assert res[0][1] == [0, 4] and res[0][2] == [100.0] * 12
    # rank 1 receives (0 + 1) * (1 + 2) = 3 records of value 1
    assert res[1][1] == [3, 0] and res[1][2] == [1.0] * 9
    assert res[0][3] is True and res[1][3] is True
    assert res[0][4] == [3.0, 20.0] and res[1][4] == [3.0, 20.0]
