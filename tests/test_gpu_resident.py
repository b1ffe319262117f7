"""GPU, resident mode: device neighbor build + device NVE + device thermo around the REBO-MoS hot
path reproduce the reference log (config #2, in.rebomos-bulk) without x/f ever visiting the host."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, POT_AEAM, POT_REBOMOS
from lammps_plugins_amd.host import capi, resident, system as S
import hostplan
import mdref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def log():
    return json.load(open(os.path.join(GOLDEN, "rebomos_bulk_log.json")))


def _domain(s, sort, v0=None, skin=2.0):
    ctx = capi.Context(0)
    p = capi.read_rebomos_file(POT_REBOMOS)
    ctx.rebomos_set_params(p)
    cutghost = 3.0 * p.rcmax[0][0] + skin
    if sort:
        d = hostplan.make_domain(ctx, capi.STYLE_REBOMOS, s, cutghost, skin, [0, 0, 1], v0=v0)
    else:   # also build the LAMMPS-style 13.4 A full list, for its statistics (log.rebomos-bulk.1:82)
        d = hostplan.Domain.single(ctx, capi.STYLE_REBOMOS, s, cutghost, skin, map_=[0, 0, 1], v0=v0, sort=False,
                                   master_list=True)
    d.cutghost = cutghost
    return ctx, d


@pytest.mark.parametrize("sort", [False, True])
def test_in_rebomos_bulk_on_device(log, sort):
    s = S.rebomos_bulk_cell()
    ctx, d = _domain(s, sort)
    assert d.nghost == log["nghost"]
    d.build_neighbors()
    st = ctx.md_neighbor_stats()
    if not sort:
        assert st[0] == log["full_neighbors"]       # FullNghs 142848 (log.rebomos-bulk.1:82)
    d.compute(eflag=1, vflag=1)
    rows = [d.thermo()]
    for step in range(1, 21):
        ev = 1 if step % 10 == 0 else 0
        d.step(eflag=ev, vflag=ev)
        if ev:
            rows.append(d.thermo())
    for got, ref in zip(rows, log["thermo"]):
        assert got["pe"] == pytest.approx(ref["pe"], abs=5.1e-5)
        assert got["ke"] == pytest.approx(ref["ke"], abs=5.1e-8)
        assert got["temp"] == pytest.approx(ref["temp"], abs=5.1e-5)
        assert got["press"] == pytest.approx(ref["press"], abs=5.1e-3)
    assert not d.needs_rebuild()                     # "Neighbor list builds = 0" (log.rebomos-bulk.1:83)
    ctx.close()


def test_device_list_forces_match_oracle(oracle):
    """device-built list (different neighbor order than the host builder) gives the oracle's forces"""
    P = oracle.rebomos_params(POT_REBOMOS)
    s = S.jitter(S.scale(S.rebomos_bulk_cell(), 1.12), 0.15, seed=1234)
    ctx, d = _domain(s, sort=False)
    d.build_neighbors()
    d.compute(eflag=3, vflag=1)
    t = d.thermo()
    got = ctx.md_download(s.n, want=("x", "f", "eatom"))
    o = mdref.RebomosCPU(oracle, P, S.System(s.box, got["x"], s.type, s.tag, s.mass)).compute(got["x"])
    assert np.abs(got["f"] - o["f_owned"]).max() < 1e-9
    assert np.abs(got["eatom"] - o["eatom_owned"]).max() < 1e-9
    assert t["pe"] == pytest.approx(o["eng"], rel=1e-10)
    assert np.allclose(t["virial"], o["virial_fdotr"], rtol=1e-9, atol=1e-7)
    ctx.close()


def test_rebuild_after_motion_keeps_energy_conserved():
    """hot start with a thin skin (0.3 A): atoms cross skin/2 every few dozen steps; rebuild on the
    device when `check yes` fires and check NVE energy conservation across rebuilds"""
    s = S.replicate(S.rebomos_bulk_cell(), (2, 1, 1))
    v0 = S.gaussian_velocities(s, 900.0, seed=4)
    ctx, d = _domain(s, sort=True, v0=v0, skin=0.3)
    d.natoms_total = s.n
    d0 = d
    d.build_neighbors()
    d.compute(eflag=1, vflag=1)
    t0 = d.thermo()
    e0 = t0["pe"] + t0["ke"]
    builds0 = d.builds
    for step in range(1, 401):
        check = step % 5 == 0
        d.ctx.md_initial_integrate()
        if check and d.needs_rebuild():
            d = hostplan.reneighbor(d, s, d0.cutghost, [0, 0, 1])   # re-wrap, re-derive ghosts, rebuild
        d.ctx.md_compute(1 if check else 0, 0)
        d.ctx.md_final_integrate()
    t1 = d.thermo()
    assert d.builds > builds0                      # the rebuild path really ran
    assert abs((t1["pe"] + t1["ke"]) - e0) / s.n < 2e-5
    ctx.close()


def _forces(s, env):
    """forces, energy and list shape of one resident compute under the given environment"""
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ctx, d = _domain(s, True)
        d.build_neighbors()
        d.compute(eflag=3, vflag=1)
        th = d.thermo()
        got = ctx.md_download(d.nlocal, want=("f", "eatom"))
        order = np.argsort(np.asarray(d.order_tag))   # back to tag order: the storage order depends on env
        info = ctx.rebomos_list_info()
        ctx.close()
        return got["f"][order], got["eatom"][order], th, info
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_tile_lists_against_the_per_cluster_path_and_large_union_classes():
    """Two independent Lennard-Jones paths on 62 k jittered atoms: tile lists (LDS-staged unions, 16-bit
    rows) and the per-cluster fallback (MDP_LJ_TILE=0).  A Z-order atom sequence and a lowered limit of the
    small launch class put part of the tiles into the large-union classes, which are exercised as well."""
    s = S.jitter(S.replicate(S.rebomos_bulk_cell(), (6, 6, 6)), 0.05, seed=3)
    f0, e0, th0, i0 = _forces(s, {"MDP_LJ_TILE": "0"})
    f1, e1, th1, i1 = _forces(s, {"MDP_LJ_TILE": "1"})
    f2, e2, th2, i2 = _forces(s, {"MDP_LJ_TILE": "1", "MDP_ORDER": "morton", "MDP_TILE_SMALL": "700"})
    assert i0["tiled"] == 0 and i1["tiled"] == 1 and i2["tiled"] == 1
    assert i1["union_max"] < i1["union_stride"]
    assert 0 < i2["large_tiles"] < i2["tiles"]          # both launch classes populated
    assert i1["clusters"] == (s.n + 1) // 2
    for f, e, th in ((f1, e1, th1), (f2, e2, th2)):
        assert np.abs(f - f0).max() < 1e-10
        assert np.abs(e - e0).max() < 1e-10
        assert th["pe"] == pytest.approx(th0["pe"], rel=1e-12)
        assert np.allclose(th["virial"], th0["virial"], rtol=1e-10, atol=1e-7)


def test_device_bytes_are_reported_per_context():
    """Pair::memory_usage() of a style must report ITS device memory, not every context's in the process"""
    L = capi.lib()
    L.mdp_device_bytes.restype = capi.C.c_double
    s_small = S.rebomos_bulk_cell()
    s_big = S.replicate(S.rebomos_bulk_cell(), (4, 4, 2))
    ctx1, d1 = _domain(s_small, True)
    ctx2, d2 = _domain(s_big, True)
    d1.build_neighbors()
    d2.build_neighbors()
    b1, b2, tot = ctx1.device_bytes(), ctx2.device_bytes(), float(L.mdp_device_bytes(None))
    assert 0 < b1 < b2                                 # the larger system holds more
    assert tot >= b1 + b2 - 1.0                        # and the process total covers both
    ctx1.close()
    assert ctx2.device_bytes() == b2
    ctx2.close()


@pytest.mark.parametrize("style", ["rebomos", "aeam"])
def test_deferred_final_integrate_gives_the_same_trajectory(style):
    """step(defer_final=True) leaves the final half-kick to the first kernel of the next step
    (mdp_md_final_initial_integrate: both half-kicks in one pass): positions, velocities and forces after 40 hot steps
    with reneighborings and thermo steps in between are bit-identical to the run with separate kernels (rebomos; the
    aeam angular kernels use FP64 atomics, whose order is not fixed: equal to rounding there)."""
    res = {}
    for defer in (False, True):
        ctx = capi.Context(0)
        if style == "rebomos":
            p = capi.read_rebomos_file(POT_REBOMOS)
            ctx.rebomos_set_params(p)
            s = S.replicate(S.rebomos_bulk_cell(), (3, 3, 3))
            v0 = S.gaussian_velocities(s, 900.0, seed=3)
            d = resident.DeviceDomain(ctx, capi.STYLE_REBOMOS, s, 3.0 * p.rcmax[0][0] + 2.0, 2.0, [0, 0, 1], v0=v0)
        else:
            af = capi.AeamFile(POT_AEAM)
            tabs = af.build()
            ctx.aeam_set_tables(tabs)
            s = S.fcc_cell(4.045, 7, frac_type2=0.02, seed=5)
            s.mass[1:3] = af.mass
            v0 = S.gaussian_velocities(s, 1500.0, seed=3)
            d = resident.DeviceDomain(ctx, capi.STYLE_AEAM, s, float(af.cut_table(tabs).max()) + 1.0, 1.0, None, v0=v0)
        d.compute(0, 0)
        pe = []
        for k in range(1, 41):
            ev = 1 if k % 10 == 0 else 0
            d.step(ev, 0, rebuild="auto", defer_final=defer and not ev)
            if ev:
                pe.append(d.thermo())
        d.flush()
        got = ctx.md_download(d.nlocal, want=("x", "v", "f"))
        order = np.argsort(d.tags_local)
        res[defer] = (got["x"][order], got["v"][order], got["f"][order], pe, d.builds)
        ctx.close()
    a, b = res[False], res[True]
    assert a[4] == b[4]                                    # same reneighborings ...
    if style == "aeam":
        assert a[4] >= 2                                   # ... and there were some (1 A of skin, 1500 K)
    for k in range(3):
        if style == "rebomos":
            assert np.array_equal(a[k], b[k])
        else:   # (the angular kernels add with FP64 atomics: two runs of the same steps differ in the last bits)
            assert np.abs(a[k] - b[k]).max() < 1e-7
    for ta, tb in zip(a[3], b[3]):
        assert ta["ke"] == pytest.approx(tb["ke"], rel=1e-13)   # (sums through atomics: not ordered)
        assert ta["pe"] == pytest.approx(tb["pe"], rel=1e-13)


def test_deferred_final_kick_is_completed_when_velocities_are_read():
    """A host that defers the final half-kick (mdp_md_defer_final) and then reads KE or velocities straight through
    the C-ABI gets full-step values: the library completes the kick itself, and the with_final call that opens the
    next step then applies only the initial half-kick (no double kick)."""
    p = capi.read_rebomos_file(POT_REBOMOS)
    s = S.replicate(S.rebomos_bulk_cell(), (2, 2, 1))
    v0 = S.gaussian_velocities(s, 600.0, seed=9)
    out = {}
    for defer in (False, True):
        ctx = capi.Context(0)
        ctx.rebomos_set_params(p)
        d = resident.DeviceDomain(ctx, capi.STYLE_REBOMOS, s, 3.0 * p.rcmax[0][0] + 2.0, 2.0, [0, 0, 1], v0=v0)
        d.compute(0, 0)
        for _ in range(5):
            d.step(0, 0, defer_final=defer)
        ke_mid = ctx.md_thermo()["ke"]                       # NOT d.thermo(): no flush on the Python side
        v_mid = ctx.md_download(d.nlocal, want=("v",))["v"]
        for _ in range(5):
            d.step(0, 0, defer_final=defer)                  # (Python still believes a kick is pending: with_final = 1)
        d.flush()
        got = ctx.md_download(d.nlocal, want=("x", "v"))
        out[defer] = (ke_mid, v_mid, got["x"], got["v"])
        ctx.close()
    assert out[True][0] == pytest.approx(out[False][0], rel=1e-14)
    for k in (1, 2, 3):
        assert np.array_equal(out[True][k], out[False][k])


def test_lane_per_centre_kernel_hands_fourth_neighbours_on(oracle):
    """The S centres of MoS2 take one lane each with room for three neighbours (rebo_centre3_kernel).  The lists are built
    on a slightly compressed, nearly perfect cell (S-S 3.04-3.09 A, just outside rcmax = 3.0 A: every S centre has its
    three bonds and is classified for that kernel); then the atoms are displaced by ~0.08 A inside the list skin, S-S
    pairs come inside rcmax and many S centres have four or five neighbours: in the first such compute they reach the
    general kernel's list, once the host has seen the count they are collected on a device list and taken by the
    8-lane-group kernel in the same step.  Forces and energies of every compute against the oracle; both routes must
    have been used."""
    P = oracle.rebomos_params(POT_REBOMOS)
    s0 = S.jitter(S.scale(S.replicate(S.rebomos_bulk_cell(), (3, 3, 1)), 0.97), 0.01, seed=77)
    ctx = capi.Context(0)
    ctx.rebomos_set_params(capi.read_rebomos_file(POT_REBOMOS))
    p = capi.read_rebomos_file(POT_REBOMOS)
    d = resident.DeviceDomain(ctx, capi.STYLE_REBOMOS, s0, 3.0 * p.rcmax[0][0] + 2.0, 2.0, [0, 0, 1])
    seen_list_mode, seen_general = False, False
    rng = np.random.default_rng(5)
    for it in range(5):
        d.compute(3, 1)
        th = d.thermo()
        got = ctx.md_download(d.nlocal, want=("x", "f", "eatom"))
        st = ctx.md_list_state()
        order = np.argsort(d.tags_local)
        x = got["x"][order]
        o = mdref.RebomosCPU(oracle, P, S.System(s0.box, x, s0.type, s0.tag, s0.mass)).compute(x)
        assert np.abs(got["f"][order] - o["f_owned"]).max() < 1e-9
        assert np.abs(got["eatom"][order] - o["eatom_owned"]).max() < 1e-9
        assert th["pe"] == pytest.approx(o["eng"], rel=1e-10)
        if it > 1:
            assert st["centre3_overflow"] > 0           # S centres with a fourth neighbour exist ...
            seen_list_mode = seen_list_mode or st["centre3_list_mode"]
        seen_general = seen_general or not st["centre3_list_mode"]
        if it == 0:
            x_built = got["x"].copy()
        # displace the atoms inside the skin (same lists: no atom moves 0.5 A), other neighbour counts every time
        ctx.md_upload_x(x_built + 0.08 * np.clip(rng.standard_normal(x_built.shape), -2.5, 2.5))
    assert seen_general and seen_list_mode              # ... and went both ways
    ctx.close()
