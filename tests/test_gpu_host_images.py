"""GPU: host mode on one periodic rank with the images kept by the library (mdp_set_box_host).

A LAMMPS host on one rank gives every ghost its owner's position plus whole box vectors (Comm::forward_comm with pbc
flags), copies fp to the images (pair_aeam.cpp:307, 946-963) and folds what the images collected back (reverse_comm).
With the host's box the library does the three on the device and moves the owned atoms' data only.  Results must be the
ones of the upload path (same oracle, same tolerances), the ghost part of x must not be read, a changing box (fix npt,
deform) must be followed step by step, and anything that is not an image of an owned atom must switch the path off."""
import numpy as np
import pytest

from conftest import POT_AEAM, POT_REBOMOS
from lammps_plugins_amd.host import capi, system as S
import mdref
import oracle_bindings as ob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P(oracle):
    return oracle.rebomos_params(POT_REBOMOS)


@pytest.fixture()
def rctx(P):
    c = capi.Context(0)
    c.rebomos_set_params(ob.product_rebomos_params(P))
    yield c
    c.close()


def _poison_ghosts(xa, nlocal):
    xb = xa.copy()
    xb[nlocal:] = np.nan                                    # the images' positions are not the host's to give any more
    return xb


@pytest.mark.parametrize("rep", [(1, 1, 1), (3, 2, 1)])
def test_rebomos_images_follow_their_owners_on_the_device(rctx, oracle, P, rep):
    """triclinic MoS2 cell (images across one, two and three box vectors): list build with uploaded images, then
    force calls with owned positions only"""
    s = S.jitter(S.replicate(S.rebomos_bulk_cell(), rep), 0.03, seed=5)
    eng = mdref.RebomosCPU(oracle, P, s, skin=2.0)
    xa = eng.all_positions(s.x)
    rctx.set_box_host(s.box)
    rctx.set_atoms_host(eng.nlocal, xa, eng.type_all, eng.tag_all, 2, map_=[0, 0, 1])
    assert rctx.host_ghosts_derived()
    rctx.set_skin(2.0)
    g = rctx.rebomos_compute_host(eng.nlocal)
    o = eng.compute(s.x)
    assert np.abs(g["f"] - o["f_owned"]).max() < 1e-9 and g["eng"] == pytest.approx(o["eng"], rel=1e-10)
    rng = np.random.default_rng(6)
    x2 = s.x.copy()
    for _ in range(3):
        x2 = x2 + 0.03 * rng.standard_normal(x2.shape)
        rctx.set_positions_host(_poison_ghosts(eng.all_positions(x2), eng.nlocal))
        g = rctx.rebomos_compute_host(eng.nlocal)
        o = eng.compute(x2)
        assert np.abs(g["f"] - o["f_owned"]).max() < 1e-9
        assert g["eng"] == pytest.approx(o["eng"], rel=1e-10)
        assert np.abs(g["eatom"] - o["eatom_owned"]).max() < 1e-9
        assert np.allclose(g["virial"], o["virial_fdotr"], rtol=1e-9, atol=1e-7)


def test_rebomos_images_follow_a_changing_box(rctx, oracle, P):
    """fix npt / deform between two list builds: atoms and box scale together, the images move by the NEW box vectors"""
    s = S.jitter(S.replicate(S.rebomos_bulk_cell(), (2, 2, 1)), 0.02, seed=7)
    eng = mdref.RebomosCPU(oracle, P, s, skin=2.0)
    rctx.set_box_host(s.box)
    rctx.set_atoms_host(eng.nlocal, eng.all_positions(s.x), eng.type_all, eng.tag_all, 2, map_=[0, 0, 1])
    assert rctx.host_ghosts_derived()
    rctx.set_skin(2.0)
    rctx.rebomos_compute_host(eng.nlocal)
    for f in (0.995, 1.004):
        s2 = S.scale(s, f)
        eng2 = mdref.RebomosCPU(oracle, P, s2, skin=2.0)
        rctx.set_box_host(s2.box)
        rctx.set_positions_host(_poison_ghosts(eng.all_positions(s.x) * f, eng.nlocal))
        g = rctx.rebomos_compute_host(eng.nlocal)
        o = eng2.compute(s2.x)
        assert np.abs(g["f"] - o["f_owned"]).max() < 1e-9
        assert g["eng"] == pytest.approx(o["eng"], rel=1e-10)


def test_rebomos_style_lists_rebuilt_under_a_changed_box(rctx, oracle, P):
    """The box changes by far more than the mirror-slot matching tolerance (count * dh = 0.15 A against 1e-3 A) and THEN
    atoms move beyond half the inner skin, so the style rebuilds its own lists -- reverse slots of image pairs included
    -- before the host hands over atoms again.  The mirror slot of (a, image of o) has to be found with the image's shift
    of THIS step, or the image clusters' forces on the owned atoms next to the periodic faces are silently dropped."""
    import dataclasses
    s = S.jitter(S.replicate(S.rebomos_bulk_cell(), (2, 2, 1)), 0.02, seed=11)
    eng = mdref.RebomosCPU(oracle, P, s, skin=2.0)
    rctx.set_box_host(s.box)
    rctx.set_atoms_host(eng.nlocal, eng.all_positions(s.x), eng.type_all, eng.tag_all, 2, map_=[0, 0, 1])
    assert rctx.host_ghosts_derived()
    rctx.set_skin(2.0)
    rctx.rebomos_compute_host(eng.nlocal)
    builds0 = rctx.rebomos_list_info()["builds"]
    for f, sign in ((1.004, 1.0), (0.996, -1.0)):
        s2 = S.scale(s, f)
        x3 = s2.x.copy()
        x3[[3, 101, 200, 555]] += sign * np.array([0.4, 0.3, 0.2])  # 0.54 A > half the 1.0 A inner skin: the style rebuilds
        s3 = dataclasses.replace(s2, x=x3)
        eng3 = mdref.RebomosCPU(oracle, P, s3, skin=2.0)
        rctx.set_box_host(s2.box)
        xa = eng.all_positions(s.x) * f
        xa[:eng.nlocal] = x3
        rctx.set_positions_host(_poison_ghosts(xa, eng.nlocal))
        g = rctx.rebomos_compute_host(eng.nlocal)
        assert rctx.rebomos_list_info()["builds"] > builds0
        builds0 = rctx.rebomos_list_info()["builds"]
        o = eng3.compute(s3.x)
        assert np.abs(g["f"] - o["f_owned"]).max() < 1e-9
        assert g["eng"] == pytest.approx(o["eng"], rel=1e-10)
        assert np.abs(g["eatom"] - o["eatom_owned"]).max() < 1e-9


def test_a_ghost_that_is_nobodys_image_switches_the_path_off(rctx, oracle, P):
    """(a) a ghost of another rank's atom (its tag is not owned here), (b) a box that is not the one the images were
    made with, (c) no box at all, (d) MDP_HOST_GHOSTS=upload: positions of all atoms are read as before"""
    s = S.jitter(S.replicate(S.rebomos_bulk_cell(), (2, 1, 1)), 0.03, seed=8)
    eng = mdref.RebomosCPU(oracle, P, s, skin=2.0)
    xa = eng.all_positions(s.x)
    o = eng.compute(s.x)
    tag_foreign = eng.tag_all.copy()
    tag_foreign[-1] = s.n + 17
    wrong_box = S.scale(s, 1.01).box
    for box, tags in ((s.box, tag_foreign), (wrong_box, eng.tag_all), (None, eng.tag_all)):
        rctx.set_box_host(box)
        rctx.set_atoms_host(eng.nlocal, xa, eng.type_all, tags, 2, map_=[0, 0, 1])
        assert not rctx.host_ghosts_derived()
        rctx.set_skin(2.0)
        g = rctx.rebomos_compute_host(eng.nlocal)
        assert np.abs(g["f"] - o["f_owned"]).max() < 1e-9
        x2 = s.x + 0.02 * np.random.default_rng(9).standard_normal(s.x.shape)
        rctx.set_positions_host(eng.all_positions(x2))
        g = rctx.rebomos_compute_host(eng.nlocal)
        assert np.abs(g["f"] - eng.compute(x2)["f_owned"]).max() < 1e-9


def test_upload_switch(oracle, P, monkeypatch):
    monkeypatch.setenv("MDP_HOST_GHOSTS", "upload")
    c = capi.Context(0)
    c.rebomos_set_params(ob.product_rebomos_params(P))
    s = S.jitter(S.rebomos_bulk_cell(), 0.03, seed=10)
    eng = mdref.RebomosCPU(oracle, P, s, skin=2.0)
    c.set_box_host(s.box)
    c.set_atoms_host(eng.nlocal, eng.all_positions(s.x), eng.type_all, eng.tag_all, 2, map_=[0, 0, 1])
    assert not c.host_ghosts_derived()
    c.close()


@pytest.mark.parametrize("device_lists", [True, False])
@pytest.mark.parametrize("ncell,frac", [(5, 0.08), (4, 0.5)])
def test_aeam_fp_and_image_forces_stay_on_the_device(oracle, ncell, frac, device_lists):
    """density half without fp / rho coming back, force half without fp going up; what the angular terms put on images
    arrives on their owners; per-atom virial likewise.  Both list sources (device-built, host CSR)."""
    T = oracle.aeam_pot(POT_AEAM)
    af = capi.AeamFile(POT_AEAM)
    ctx = capi.Context(0)
    ctx.aeam_set_tables(af.build())
    s = S.jitter(S.fcc_cell(4.045, ncell, frac_type2=frac, seed=99), 0.06, seed=100)
    eng = mdref.AeamCPU(oracle, T, s)
    xa = eng.all_positions(s.x)
    nall, n = len(xa), eng.nlocal
    ctx.set_box_host(s.box)
    if device_lists:
        ctx.aeam_device_lists(True)
    ctx.set_atoms_host(n, xa, eng.type_all, eng.tag_all, 2, map_=None)
    assert ctx.host_ghosts_derived()
    if device_lists:
        ctx.set_skin(1.0)
    else:
        ctx.set_neighbors_csr_host(eng.nn, eng.off, eng.nb, 1.0)

    def both(x, eflag, vflag):
        d = ctx.aeam_density_host(n, eflag=eflag, keep_fp=True)
        r = ctx.aeam_force_host(nall, n, None, eflag=eflag, vflag=vflag)
        assert not r["f"][n:].any()                          # nothing for the host's reverse_comm to carry
        return d, r

    d, r = both(s.x, 3, 5)
    o = eng.compute(s.x)
    assert np.abs(r["f"][:n] - o["f_owned"]).max() < 1e-9
    assert d["eng"] + r["eng"] == pytest.approx(o["eng"], rel=1e-11)
    assert np.abs(d["eatom"] + r["eatom"] - o["eatom"][:n]).max() < 1e-9
    assert np.allclose(r["virial"], o["virial_fdotr"], rtol=1e-9, atol=1e-7)
    vo = o["vatom"][:n].copy()
    np.add.at(vo, eng.owner, o["vatom"][n:])
    assert not r["vatom"][n:].any()
    assert np.abs(r["vatom"][:n] - vo).max() < 1e-9 * max(1.0, np.abs(vo).max())
    # force-only steps between two list builds: owned positions only, no host read until the forces
    rng = np.random.default_rng(3)
    x2 = s.x.copy()
    for _ in range(2):
        x2 = x2 + 0.02 * rng.standard_normal(x2.shape)
        ctx.set_positions_host(_poison_ghosts(eng.all_positions(x2), n))
        d, r = both(x2, 0, 0)
        assert np.abs(r["f"][:n] - eng.compute(x2)["f_owned"]).max() < 1e-9
    # the two-array protocol still works on the same context (a host that wants fp back)
    d = ctx.aeam_density_host(n, eflag=0)
    r = ctx.aeam_force_host(nall, n, np.concatenate([d["fp"], d["fp"][eng.owner]]), eflag=0, vflag=0)
    assert np.abs(r["f"][:n] - eng.compute(x2)["f_owned"]).max() < 1e-9
    ctx.close()


@pytest.mark.parametrize("style", ["rebomos", "aeam"])
def test_kept_and_uploaded_images_agree_at_scale(style):
    """a few hundred thousand atoms (many tiles, Hilbert-sorted device order, images across all three box vectors): the
    same displaced positions through a context that keeps the images and one that takes them from the host"""
    rng = np.random.default_rng(11)
    if style == "rebomos":
        s = S.jitter(S.replicate(S.rebomos_bulk_cell(), (10, 10, 10)), 0.02, seed=12)     # 288 k atoms
        p = capi.read_rebomos_file(POT_REBOMOS)
        cut, skin, map_ = 3.0 * p.rcmax[0][0] + 2.0, 2.0, [0, 0, 1]
    else:
        s = S.jitter(S.fcc_cell(4.045, 40, frac_type2=0.0075, seed=13), 0.05, seed=14)    # 256 k atoms
        af = capi.AeamFile(POT_AEAM)
        tabs = af.build()
        cut, skin, map_ = float(af.cut_table(tabs).max()) + 1.0, 1.0, None
    xa, ta, ga, owner, shift, n, ng = S.with_ghosts(s, cut)
    nall = n + ng
    out = []
    for box in (s.box, None):
        ctx = capi.Context(0)
        if style == "rebomos":
            ctx.rebomos_set_params(p)
        else:
            ctx.aeam_set_tables(tabs)
            ctx.aeam_device_lists(True)
        ctx.set_box_host(box)
        ctx.set_atoms_host(n, xa, ta, ga, 2, map_=map_)
        assert ctx.host_ghosts_derived() == (box is not None)
        ctx.set_skin(skin)
        disp = np.zeros_like(xa)
        rng2 = np.random.default_rng(15)
        f = None
        for step in range(3):
            if step:
                disp[:n] += 0.03 * rng2.standard_normal((n, 3))
                disp[n:] = disp[owner]
                xnew = xa + disp
                ctx.set_positions_host(_poison_ghosts(xnew, n) if box is not None else xnew)
            if style == "rebomos":
                f = ctx.rebomos_compute_host(n, eflag=0, vflag=0)["f"]
            else:
                keep = box is not None
                d = ctx.aeam_density_host(n, eflag=0, keep_fp=keep)
                fp_all = None if keep else np.concatenate([d["fp"], d["fp"][owner]])
                r = ctx.aeam_force_host(nall, n, fp_all, eflag=0, vflag=0)
                f = r["f"][:n].copy()
                if not keep:
                    np.add.at(f, owner, r["f"][n:])
        out.append(f)
        ctx.close()
    assert np.abs(out[0]).max() > 0.1
    assert np.abs(out[0] - out[1]).max() < 1e-10


def test_box_arguments(rctx):
    """a box with a non-positive length is refused; NULL withdraws the box and with it the derived images"""
    h = np.array([10.0, 0.0, 10.0, 0.0, 0.0, 0.0])
    assert rctx.L.mdp_set_box_host(rctx.h, capi._dp(h)) != 0
    assert "positive" in rctx.L.mdp_last_error(rctx.h).decode()
    assert rctx.L.mdp_set_box_host(rctx.h, None) == 0
    assert not rctx.host_ghosts_derived()
    assert rctx.L.mdp_host_ghosts_derived(None) == 0
