"""GPU: the device-side domain decomposition (csrc/domain.hip) -- remap, migration, Hilbert ordering, ghost
derivation, per-step halo -- against the reference's published counts (log.rebomos-bulk.1:72-75, .4:72-75),
against the host-planned numpy decomposition (host/decomp.py, independent code) and against the CPU oracle.
Several ranks run as threads of this one process (resident.ThreadTransport): a GPU box admits few processes
on its card, and the rank code is the same that runs over RCCL."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, POT_AEAM, POT_REBOMOS
from lammps_plugins_amd.host import capi, decomp, resident, system as S
import mdref

pytestmark = pytest.mark.gpu
MAP = [0, 0, 1]


@pytest.fixture(scope="module")
def log():
    return json.load(open(os.path.join(GOLDEN, "rebomos_bulk_log.json")))


def _rebo_ctx():
    ctx = capi.Context(0)
    p = capi.read_rebomos_file(POT_REBOMOS)
    ctx.rebomos_set_params(p)
    return ctx, 3.0 * p.rcmax[0][0] + 2.0


def _by_tag(dom, want=("x", "v", "f")):
    got = dom.ctx.md_download(dom.nlocal, want=want)
    tags = dom.tags_local
    return tags, {k: got[k] for k in want}


def test_one_rank_reproduces_the_reference_log(log):
    """288 atoms, 4285 ghosts (log.rebomos-bulk.1:72-75), thermo rows of steps 0/10/20 (:54-56)"""
    s = S.rebomos_bulk_cell()
    ctx, cutghost = _rebo_ctx()
    d = resident.DeviceDomain(ctx, capi.STYLE_REBOMOS, s, cutghost, 2.0, MAP)
    assert (d.nlocal, d.nself, d.nrecv) == (288, log["nghost"], 0)
    assert sorted(d.tags_local.tolist()) == list(range(1, 289))
    d.compute(1, 1)
    rows = [d.thermo()]
    for step in range(1, 21):
        ev = 1 if step % 10 == 0 else 0
        d.step(ev, ev)
        if ev:
            rows.append(d.thermo())
    for got, ref in zip(rows, log["thermo"]):
        assert got["pe"] == pytest.approx(ref["pe"], abs=5.1e-5)
        assert got["ke"] == pytest.approx(ref["ke"], abs=5.1e-8)
        assert got["press"] == pytest.approx(ref["press"], abs=5.1e-3)
    assert not d.moved()                     # "Neighbor list builds = 0" (log.rebomos-bulk.1:83)
    ctx.close()


def test_device_ghosts_equal_the_host_planned_ghosts():
    """same ghost SET (tag, position) as system.make_ghosts for a strained, jittered triclinic cell"""
    s = S.jitter(S.scale(S.replicate(S.rebomos_bulk_cell(), (2, 1, 2)), 1.05), 0.1, seed=8)
    ctx, cutghost = _rebo_ctx()
    d = resident.DeviceDomain(ctx, capi.STYLE_REBOMOS, s, cutghost, 2.0, MAP)
    owner, shift = S.make_ghosts(s.box, S.wrap(s.box, s.x), cutghost)
    assert d.nself == len(owner)
    want = np.round(S.wrap(s.box, s.x)[owner] + S.mul_upper(shift, s.box.h), 6)
    g = np.round(ctx.md_download_x_all(d.nlocal + d.nghost)[d.nlocal:], 6)
    assert sorted(map(tuple, g)) == sorted(map(tuple, want))
    ctx.close()


def test_atoms_outside_the_box_are_remapped(oracle):
    """input positions shifted by whole and fractional box vectors: the device wraps them (Domain::remap) and
    the forces equal the oracle's for the wrapped system"""
    P = oracle.rebomos_params(POT_REBOMOS)
    s = S.jitter(S.scale(S.rebomos_bulk_cell(), 1.08), 0.12, seed=21)
    ctx, cutghost = _rebo_ctx()
    d = resident.DeviceDomain(ctx, capi.STYLE_REBOMOS, s, cutghost, 2.0, MAP)
    # push atoms out through every face, then reneighbor on the device
    got = ctx.md_download(d.nlocal, want=("x",))
    tags = d.tags_local
    rng = np.random.default_rng(5)
    kick = rng.integers(-1, 2, size=(d.nlocal, 3)).astype(float) @ s.box.h.T  # (main thread only)
    ctx.md_upload_x(got["x"] + kick)
    d.reneighbor()
    assert d.nlocal == s.n and sorted(d.tags_local.tolist()) == list(range(1, s.n + 1))
    d.compute(3, 1)
    t = d.thermo()
    tags, a = _by_tag(d, ("x", "f"))
    order = np.argsort(tags)
    xs = a["x"][order]
    lam = s.box.x2lamda(xs)
    assert lam.min() >= -1e-12 and lam.max() < 1.0 + 1e-12
    o = mdref.RebomosCPU(oracle, P, S.System(s.box, xs, s.type, s.tag, s.mass)).compute(xs)
    assert np.abs(a["f"][order] - o["f_owned"]).max() < 1e-9
    assert t["pe"] == pytest.approx(o["eng"], rel=1e-10)
    ctx.close()


def _run(world, s, v0, steps, rebuild_every, style=capi.STYLE_REBOMOS, pot=None, defer=False, self_remote=False):
    """NVE run on `world` bricks (threads), forced reneighboring every `rebuild_every` steps; returns per-tag
    x, v and the thermo of the last step, plus per-rank counts"""

    def rank_fn(r, make_tr):
        if style == capi.STYLE_REBOMOS:
            ctx, cutghost = _rebo_ctx()
            skin, map_ = 2.0, MAP
        else:
            ctx = capi.Context(0)
            af = capi.AeamFile(pot or POT_AEAM)
            tabs = af.build()
            ctx.aeam_set_tables(tabs)
            skin, map_ = 1.0, None
            cutghost = float(af.cut_table(tabs).max()) + skin
            ctx._af = (af, tabs)
        tr = make_tr(ctx) if world > 1 else None
        d = resident.DeviceDomain(ctx, style, s, cutghost, skin, map_, v0=v0, transport=tr, self_remote=self_remote)
        counts0 = (d.nlocal, d.nself, d.nrecv)
        d.compute(1, 1)
        th0 = d.thermo()
        left = 0
        for step in range(1, steps + 1):
            rb = rebuild_every and step % rebuild_every == 0
            d.step(1 if step == steps else 0, 1 if step == steps else 0, rebuild=rb, defer_final=defer and step != steps)
            if rb:
                left += ctx.dd_info()["left_last"]
        th = d.thermo()
        tags, a = _by_tag(d, ("x", "v", "f"))
        aeam = ctx.md_aeam_state() if style == capi.STYLE_AEAM else None
        prunes = ctx.md_prune_stats()
        near = None
        if aeam and world > 1 and aeam["interior_tiles"]:
            # distance from the atoms of the tiles that run before the halo arrives to the nearest remote ghost
            info = ctx.dd_info()
            xa = ctx.md_download_x_all(info["nlocal"] + info["nself"] + info["nrecv"])
            rs = info["nlocal"] + info["nself"]
            inner = xa[:min(aeam["interior_tiles"] * 32, info["nlocal"])]
            near = min(float(np.linalg.norm(xa[rs:] - p, axis=1).min()) for p in inner)
        ctx.close()
        return dict(tags=tags, x=a["x"], v=a["v"], f=a["f"], th0=th0, th=th, counts0=counts0, left=left,
                    builds=d.builds, aeam=aeam, overlapped=d.aeam_overlapped, prunes=prunes, near=near)

    if world == 1:
        res = [rank_fn(0, None)]
    else:
        res = resident.run_ranks(world, rank_fn)
    n = s.n
    x, v, f = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 3))
    seen = np.zeros(n, dtype=int)
    for r in res:
        idx = r["tags"] - 1
        x[idx], v[idx], f[idx] = r["x"], r["v"], r["f"]
        seen[idx] += 1
    assert np.all(seen == 1)                      # every atom owned exactly once
    return dict(x=x, v=v, f=f, th0=res[0]["th0"], th=res[0]["th"], counts0=[r["counts0"] for r in res],
                left=sum(r["left"] for r in res), builds=res[0]["builds"], ranks=res)


def test_four_bricks_match_the_reference_4_rank_log(log):
    """2 x 2 x 1 bricks of the 288-atom cell: Nlocal 72 each, Nghost 2768/2768/2775/2775
    (log.rebomos-bulk.4:72-75), and the same thermo rows as one rank (log.rebomos-bulk.4:54-56 == .1:54-56)"""
    s = S.rebomos_bulk_cell()
    r = _run(4, s, None, 20, 0)
    assert [c[0] for c in r["counts0"]] == [72, 72, 72, 72]
    assert sorted(c[1] + c[2] for c in r["counts0"]) == [2768, 2768, 2775, 2775]
    ref0, ref20 = log["thermo"][0], log["thermo"][2]
    assert r["th0"]["pe"] == pytest.approx(ref0["pe"], abs=5.1e-5)
    assert r["th0"]["press"] == pytest.approx(ref0["press"], abs=5.1e-3)
    assert r["th"]["pe"] == pytest.approx(ref20["pe"], abs=5.1e-5)
    assert r["th"]["ke"] == pytest.approx(ref20["ke"], abs=5.1e-8)
    assert r["th"]["press"] == pytest.approx(ref20["press"], abs=5.1e-3)


@pytest.mark.parametrize("world", [2, 3, 6, 8])
def test_bricks_with_migration_follow_the_one_rank_trajectory(world):
    """3x3x2 replica (5184 atoms) at 300 K with a uniform drift of 60 A/ps on top: atoms stream through brick
    and box faces (Galilean invariance: the physics is that of the run at rest).  Reneighboring is forced every
    2 steps (0.16 A of drift, far inside the 1 A half skin).  1, 2 and 8 bricks must produce the same
    trajectory, and atoms must really have changed owner."""
    s = S.replicate(S.rebomos_bulk_cell(), (3, 3, 2))
    v0 = S.gaussian_velocities(s, 300.0, seed=11) + np.array([60.0, -45.0, 30.0])
    steps, every = 40, 2
    one = _run(1, s, v0, steps, every)
    many = _run(world, s, v0, steps, every)
    assert many["left"] > 20                       # atoms migrated between bricks
    assert many["builds"] == one["builds"] == 1 + steps // every
    dx = many["x"] - one["x"]
    dx -= np.round(s.box.x2lamda(dx + s.box.lo)) @ s.box.h.T      # same atom, possibly another periodic image
    assert np.abs(dx).max() < 1e-8
    assert np.abs(many["v"] - one["v"]).max() < 1e-7
    assert np.abs(many["f"] - one["f"]).max() < 1e-7
    assert many["th"]["pe"] == pytest.approx(one["th"]["pe"], rel=1e-10)
    assert many["th"]["ke"] == pytest.approx(one["th"]["ke"], rel=1e-9)
    assert np.allclose(many["th"]["virial"], one["th"]["virial"], rtol=1e-8, atol=1e-5)
    # and the drifting run is the run at rest, seen from a moving frame
    rest = _run(1, s, v0 - np.array([60.0, -45.0, 30.0]), steps, 0)
    assert many["th"]["pe"] == pytest.approx(rest["th"]["pe"], rel=1e-9)


@pytest.mark.parametrize("world", [2, 4])
def test_strained_hot_bricks_queue_their_cubic_pairs(world, oracle):
    """the x 1.12 cell with jitter (pairs on the cubic inner Lennard-Jones spline in every tile) on bricks: after the first
    compute every rank takes the tile kernel that queues those pairs, with remote ghosts among the queued neighbours;
    forces equal the oracle's, the trajectory the one-brick run's"""
    P = oracle.rebomos_params(POT_REBOMOS)
    s = S.jitter(S.scale(S.replicate(S.rebomos_bulk_cell(), (3, 3, 2)), 1.12), 0.15, seed=1234)
    v0 = S.gaussian_velocities(s, 300.0, seed=21) + np.array([40.0, -30.0, 20.0])
    xw = S.wrap(s.box, s.x)
    o = mdref.RebomosCPU(oracle, P, S.System(s.box, xw, s.type, s.tag, s.mass)).compute(xw)
    st = _run(world, s, v0, 0, 0)
    assert np.abs(st["f"] - o["f_owned"]).max() < 1e-9
    one = _run(1, s, v0, 20, 5)
    many = _run(world, s, v0, 20, 5)
    dx = many["x"] - one["x"]
    dx -= np.round(s.box.x2lamda(dx + s.box.lo)) @ s.box.h.T
    assert np.abs(dx).max() < 1e-8
    assert np.abs(many["f"] - one["f"]).max() < 1e-7
    assert many["th"]["pe"] == pytest.approx(one["th"]["pe"], rel=1e-10)
    assert np.allclose(many["th"]["virial"], one["th"]["virial"], rtol=1e-8, atol=1e-5)


def test_aeam_bricks_with_halo_of_fp_and_ghost_forces(oracle):
    """AEAM on 2 and 4 bricks: scalar forward exchange of fp and reverse exchange of the angular ghost forces
    through the library's send list; forces equal the oracle's, trajectory equals the one-rank run"""
    T = oracle.aeam_pot(POT_AEAM)
    s = S.jitter(S.fcc_cell(4.045, 8, frac_type2=0.06, seed=3), 0.05, seed=4)
    s.mass[1:3] = capi.AeamFile(POT_AEAM).mass[:2]
    v0 = S.gaussian_velocities(s, 600.0, seed=2) + np.array([40.0, 25.0, -30.0])
    r0 = _run(1, s, v0, 0, 0, style=capi.STYLE_AEAM)
    o = mdref.AeamCPU(oracle, T, S.System(s.box, S.wrap(s.box, s.x), s.type, s.tag, s.mass)).compute(S.wrap(s.box, s.x))
    assert np.abs(r0["f"] - o["f_owned"]).max() < 1e-9
    one = _run(1, s, v0, 24, 3, style=capi.STYLE_AEAM)
    for world in (2, 4):
        st = _run(world, s, v0, 0, 0, style=capi.STYLE_AEAM)
        assert np.abs(st["f"] - o["f_owned"]).max() < 1e-9
        assert st["th0"]["pe"] == pytest.approx(o["eng"], rel=1e-11)
        many = _run(world, s, v0, 24, 3, style=capi.STYLE_AEAM)
        assert many["left"] > 5
        dx = many["x"] - one["x"]
        dx -= np.round(s.box.x2lamda(dx + s.box.lo)) @ s.box.h.T
        assert np.abs(dx).max() < 1e-8
        assert many["th"]["pe"] == pytest.approx(one["th"]["pe"], rel=1e-10)


@pytest.mark.parametrize("world,temp,drift", [(2, 863.0, 0.0), (8, 600.0, 1.0), (4, 863.0, 1.0)])
def test_aeam_exchanges_behind_the_interior_tiles(oracle, world, temp, drift):
    """A system large enough for bricks with an interior (16^3 fcc cells, 16 384 atoms, 64.7 A): the shell atoms of a
    brick are stored behind the interior ones, so the tiles that reach no remote ghost are a leading range; their
    density runs while the positions travel, their pair forces while fp and the three-body forces on ghosts travel
    (mdp_md_compute_begin / mdp_md_aeam_force_begin).  Forces equal the oracle's, the trajectory -- with migrations,
    prunings that force a step back onto the blocking path, thermo steps -- equals the one-brick run."""
    T = oracle.aeam_pot(POT_AEAM)
    s = S.jitter(S.fcc_cell(4.045, 16, frac_type2=0.03, seed=5), 0.05, seed=6)
    s.mass[1:3] = capi.AeamFile(POT_AEAM).mass[:2]
    v0 = S.gaussian_velocities(s, temp, seed=7) + drift * np.array([40.0, 25.0, -30.0])
    xw = S.wrap(s.box, s.x)
    o = mdref.AeamCPU(oracle, T, S.System(s.box, xw, s.type, s.tag, s.mass)).compute(xw)
    st = _run(world, s, v0, 0, 0, style=capi.STYLE_AEAM)
    assert np.abs(st["f"] - o["f_owned"]).max() < 1e-9
    assert st["th0"]["pe"] == pytest.approx(o["eng"], rel=1e-11)
    steps, every = 30, 6
    one = _run(1, s, v0, steps, every, style=capi.STYLE_AEAM)
    many = _run(world, s, v0, steps, every, style=capi.STYLE_AEAM, defer=True)
    for r in many["ranks"]:
        a = r["aeam"]
        assert 0 < a["interior_tiles"] < a["tiles"]          # every brick has an interior and a shell
        assert a["ghost_forces"]                             # 3 % angular atoms: some sit in the shell
        assert 0 < r["overlapped"] <= steps                  # steps on the phased path ...
        assert r["near"] > 5.0                               # no remote ghost within reach of an early tile (the tile
        #                                                      that straddles the end of the interior atoms holds few)
        if drift or temp > 700:
            assert r["prunes"]["prunings"] > 1               # ... and prunings (those steps take the blocking path)
    frac = sum(r["aeam"]["interior_tiles"] for r in many["ranks"]) / sum(r["aeam"]["tiles"] for r in many["ranks"])
    assert frac > (0.4 if world == 2 else 0.08)              # (64.7 A / 2 per brick, shell 7.5 A on both sides)
    if drift:
        assert many["left"] > 20
    dx = many["x"] - one["x"]
    dx -= np.round(s.box.x2lamda(dx + s.box.lo)) @ s.box.h.T
    assert np.abs(dx).max() < 1e-8
    assert np.abs(many["v"] - one["v"]).max() < 1e-7
    assert np.abs(many["f"] - one["f"]).max() < 1e-7
    assert many["th"]["pe"] == pytest.approx(one["th"]["pe"], rel=1e-10)
    assert many["th"]["ke"] == pytest.approx(one["th"]["ke"], rel=1e-9)
    assert np.allclose(many["th"]["virial"], one["th"]["virial"], rtol=1e-8, atol=1e-5)


@pytest.mark.parametrize("world", [2, 4])
def test_aeam_pure_metal_bricks_keep_one_exchange_schedule(world):
    """No angular atom anywhere: nothing is ever put on a remote ghost and the reverse exchange is skipped by all ranks
    (`ghost_forces` false).  A rank whose rows are due for pruning takes the blocking order for that step while its
    peers stay on the phased one -- a rank-local decision -- and must still issue exactly the peers' exchanges (fp
    forward, nothing back), or the collectives of the ranks pair up wrongly.  Hot, drifting run with prunings."""
    s = S.jitter(S.fcc_cell(4.045, 16, frac_type2=0.0, seed=21), 0.05, seed=22)
    s.mass[1:3] = capi.AeamFile(POT_AEAM).mass[:2]
    v0 = S.gaussian_velocities(s, 863.0, seed=23) + np.array([40.0, 25.0, -30.0])
    steps, every = 36, 12
    one = _run(1, s, v0, steps, every, style=capi.STYLE_AEAM)
    many = _run(world, s, v0, steps, every, style=capi.STYLE_AEAM, defer=True)
    for r in many["ranks"]:
        assert not r["aeam"]["ghost_forces"]
        assert r["prunes"]["prunings"] > 1                   # prunings between the reneighborings: blocking steps ...
        assert 0 < r["overlapped"] < steps                   # ... among phased ones
    assert many["left"] > 20
    dx = many["x"] - one["x"]
    dx -= np.round(s.box.x2lamda(dx + s.box.lo)) @ s.box.h.T
    assert np.abs(dx).max() < 1e-8
    assert np.abs(many["f"] - one["f"]).max() < 1e-7
    assert many["th"]["pe"] == pytest.approx(one["th"]["pe"], rel=1e-10)
    assert np.allclose(many["th"]["virial"], one["th"]["virial"], rtol=1e-8, atol=1e-5)


def test_aeam_phases_on_one_rank_with_every_image_remote(oracle):
    """the same phases on ONE brick whose periodic images are all treated as remote ghosts (`self_remote`, thread
    transport with one rank): shell = everything within the ghost cutoff of the box faces"""
    s = S.jitter(S.fcc_cell(4.045, 12, frac_type2=0.03, seed=15), 0.05, seed=16)
    s.mass[1:3] = capi.AeamFile(POT_AEAM).mass[:2]
    v0 = S.gaussian_velocities(s, 863.0, seed=17)
    one = _run(1, s, v0, 20, 5, style=capi.STYLE_AEAM)

    def rank_fn(r, make_tr):
        ctx = capi.Context(0)
        af = capi.AeamFile(POT_AEAM)
        tabs = af.build()
        ctx.aeam_set_tables(tabs)
        ctx._af = (af, tabs)
        cutghost = float(af.cut_table(tabs).max()) + 1.0
        d = resident.DeviceDomain(ctx, capi.STYLE_AEAM, s, cutghost, 1.0, None, v0=v0, transport=make_tr(ctx),
                                  self_remote=True)
        assert d.nself == 0 and d.nrecv > 0
        d.compute(1, 1)
        for step in range(1, 21):
            d.step(1 if step == 20 else 0, 1 if step == 20 else 0, rebuild=step % 5 == 0, defer_final=step != 20)
        th = d.thermo()
        tags, a = _by_tag(d, ("x", "f"))
        state, n = ctx.md_aeam_state(), d.aeam_overlapped
        ctx.close()
        return tags, a, th, state, n

    (tags, a, th, state, n), = resident.run_ranks(1, rank_fn)
    assert 0 < state["interior_tiles"] < state["tiles"] and n > 0
    order = np.argsort(tags)
    dx = a["x"][order] - one["x"]
    dx -= np.round(s.box.x2lamda(dx + s.box.lo)) @ s.box.h.T
    assert np.abs(dx).max() < 1e-8
    assert np.abs(a["f"][order] - one["f"]).max() < 1e-7
    assert th["pe"] == pytest.approx(one["th"]["pe"], rel=1e-10)
    assert np.allclose(th["virial"], one["th"]["virial"], rtol=1e-8, atol=1e-5)


def test_aeam_bricks_with_five_atom_types(oracle, tmp_path):
    """the same on 2 and 4 bricks with a five-element file (tile kernels with per-entry types; types of remote ghosts
    arrive with the border exchange), the steps run with the final half-kick deferred into the next step's kernel"""
    import aeam_five
    path = str(tmp_path / "five.aeam")
    aeam_five.write_five_element_file(path, POT_AEAM)
    af5 = capi.AeamFile(path)
    T5 = oracle.aeam_pot(path)
    s2 = S.jitter(S.fcc_cell(4.045, 8, frac_type2=0.06, seed=3), 0.05, seed=4)
    rng = np.random.default_rng(11)
    t5 = np.where(s2.type == 1, rng.integers(1, 4, s2.n), rng.integers(4, 6, s2.n)).astype(np.int32)
    s = S.System(s2.box, s2.x.copy(), t5, s2.tag.copy(), np.array([0.0] + list(af5.mass)))
    v0 = S.gaussian_velocities(s, 600.0, seed=2) + np.array([40.0, 25.0, -30.0])
    xw = S.wrap(s.box, s.x)
    o = mdref.AeamCPU(oracle, T5, S.System(s.box, xw, s.type, s.tag, s.mass)).compute(xw)
    one = _run(1, s, v0, 24, 3, style=capi.STYLE_AEAM, pot=path)
    for world in (2, 4):
        st = _run(world, s, v0, 0, 0, style=capi.STYLE_AEAM, pot=path)
        assert np.abs(st["f"] - o["f_owned"]).max() < 1e-9
        assert st["th0"]["pe"] == pytest.approx(o["eng"], rel=1e-11)
        many = _run(world, s, v0, 24, 3, style=capi.STYLE_AEAM, pot=path, defer=True)
        assert many["left"] > 5
        dx = many["x"] - one["x"]
        dx -= np.round(s.box.x2lamda(dx + s.box.lo)) @ s.box.h.T
        assert np.abs(dx).max() < 1e-8
        assert np.abs(many["v"] - one["v"]).max() < 1e-7
        assert many["th"]["pe"] == pytest.approx(one["th"]["pe"], rel=1e-10)


def test_aeam_twelve_atom_types_hot_on_two_bricks(oracle, tmp_path):
    """more atom types than the kernel-argument parameter block holds (12 > 8): the generic kernels on one and on two
    bricks (types of remote ghosts arrive with the border exchange), 24 hot steps with migrations; forces against the
    oracle, trajectory against the one-brick run"""
    import aeam_five
    path = str(tmp_path / "twelve.aeam")
    cls = [0] * 7 + [1] * 5
    aeam_five.write_relabelled_file(path, POT_AEAM, cls, ["M%d" % k for k in range(7)] + ["X%d" % k for k in range(5)])
    af = capi.AeamFile(path)
    assert af.nelements == 12 and af.nnonangular == 7
    T = oracle.aeam_pot(path)
    s2 = S.jitter(S.fcc_cell(4.045, 8, frac_type2=0.06, seed=3), 0.05, seed=4)
    rng = np.random.default_rng(12)
    tn = np.where(s2.type == 1, rng.integers(1, 8, s2.n), rng.integers(8, 13, s2.n)).astype(np.int32)
    assert len(set(tn.tolist())) == 12
    s = S.System(s2.box, s2.x.copy(), tn, s2.tag.copy(), np.array([0.0] + list(af.mass)))
    v0 = S.gaussian_velocities(s, 600.0, seed=2) + np.array([40.0, 25.0, -30.0])
    xw = S.wrap(s.box, s.x)
    o = mdref.AeamCPU(oracle, T, S.System(s.box, xw, s.type, s.tag, s.mass)).compute(xw)
    one = _run(1, s, v0, 24, 3, style=capi.STYLE_AEAM, pot=path)
    st = _run(2, s, v0, 0, 0, style=capi.STYLE_AEAM, pot=path)
    assert np.abs(st["f"] - o["f_owned"]).max() < 1e-9
    assert st["th0"]["pe"] == pytest.approx(o["eng"], rel=1e-11)
    many = _run(2, s, v0, 24, 3, style=capi.STYLE_AEAM, pot=path, defer=True)
    assert many["left"] > 5
    dx = many["x"] - one["x"]
    dx -= np.round(s.box.x2lamda(dx + s.box.lo)) @ s.box.h.T
    assert np.abs(dx).max() < 1e-8
    assert many["th"]["pe"] == pytest.approx(one["th"]["pe"], rel=1e-10)


def test_deferred_displacement_trigger_fires_once_per_need():
    """mdp_md_moved_async: silent while atoms stay inside skin/2 - margin, fires one call after they leave it,
    and is reset by the reneighboring it asks for"""
    s = S.replicate(S.rebomos_bulk_cell(), (2, 2, 1))
    ctx, cutghost = _rebo_ctx()
    v0 = np.tile(np.array([95.0, 0.0, 0.0]), (s.n, 1))           # 0.095 A per step
    d = resident.DeviceDomain(ctx, capi.STYLE_REBOMOS, s, cutghost, 2.0, MAP, v0=v0)
    d.compute(0, 0)
    fired = []
    for step in range(1, 31):
        ctx.md_initial_integrate()
        m = d.moved()
        if m:
            d.reneighbor()
            fired.append(step)
        ctx.md_compute(0, 0)
        ctx.md_final_integrate()
    # trigger distance 0.9 A: exceeded at step 10 (0.95 A), reported one call later, then again 11 steps on
    assert fired == [11, 22]
    assert d.dangerous == 0
    ctx.close()


@pytest.mark.parametrize("style", ["rebomos", "aeam"])
def test_rccl_transport_inside_the_library_on_one_gpu(style, oracle):
    """csrc/comm_rccl.hip with a one-rank communicator: `self_remote` sends every periodic self-image through the
    transport to the rank itself, so border records, the per-step position exchange (on its own stream, overlapped
    with the interior tiles), AEAM's fp forward and force reverse exchanges and the all-reduce all run through
    ncclSend / ncclRecv / ncclAllGather / ncclAllReduce on this GPU.  Results must equal the plain one-GPU run."""
    if style == "rebomos":
        s = S.jitter(S.replicate(S.rebomos_bulk_cell(), (2, 2, 1)), 0.05, seed=17)
        v0 = S.gaussian_velocities(s, 300.0, seed=3) + np.array([80.0, -30.0, 20.0])
        st = capi.STYLE_REBOMOS
    else:
        s = S.jitter(S.fcc_cell(4.045, 12, frac_type2=0.05, seed=9), 0.04, seed=10)   # 48.5 A: a brick with an interior
        s.mass[1:3] = capi.AeamFile(POT_AEAM).mass[:2]
        v0 = S.gaussian_velocities(s, 600.0, seed=4) + np.array([50.0, 35.0, -40.0])
        st = capi.STYLE_AEAM

    def run(native):
        if st == capi.STYLE_REBOMOS:
            ctx, cutghost = _rebo_ctx()
            skin, map_ = 2.0, MAP
        else:
            ctx = capi.Context(0)
            af = capi.AeamFile(POT_AEAM)
            tabs = af.build()
            ctx.aeam_set_tables(tabs)
            ctx._af = (af, tabs)
            skin, map_ = 1.0, None
            cutghost = float(af.cut_table(tabs).max()) + skin
        tr = resident.NativeTransport(1, 0) if native else None
        d = resident.DeviceDomain(ctx, st, s, cutghost, skin, map_, v0=v0, transport=tr, self_remote=native)
        if native:
            assert d.nself == 0 and d.nrecv > 0 and d.nsend == d.nrecv     # every ghost travels
        d.compute(1, 1)
        th0 = d.thermo()
        for step in range(1, 21):
            d.step(0, 0, rebuild=step % 5 == 0)
        d.compute(1, 1)
        th = d.thermo()
        tags, a = _by_tag(d, ("x", "f"))
        order = np.argsort(tags)
        ghosts = d.nself + d.nrecv
        if native and st == capi.STYLE_AEAM:   # fp and ghost forces travelled in one group behind the interior force tiles
            assert d.aeam_overlapped > 0 and 0 < ctx.md_aeam_state()["interior_tiles"]
        ctx.close()
        return th0, th, a["x"][order], a["f"][order], ghosts

    p0, p1, px, pf, pg = run(False)
    n0, n1, nx, nf, ng = run(True)
    assert ng == pg
    assert n0["pe"] == pytest.approx(p0["pe"], rel=1e-12)
    assert np.allclose(n0["virial"], p0["virial"], rtol=1e-10, atol=1e-7)
    assert n1["pe"] == pytest.approx(p1["pe"], rel=1e-10)
    assert n1["ke"] == pytest.approx(p1["ke"], rel=1e-9)
    dx = nx - px
    dx -= np.round(s.box.x2lamda(dx + s.box.lo)) @ s.box.h.T
    assert np.abs(dx).max() < 1e-9
    assert np.abs(nf - pf).max() < 1e-7


@pytest.mark.parametrize("style", ["rebomos", "aeam"])
def test_whole_steps_in_the_library_with_the_check_flag_riding_in_the_halo(style):
    """mdp_dd_comm_step_begin / _end (one-rank communicator, every image remote): integrate, the `check yes` decision
    from the word gathered behind the PREVIOUS step's position exchange, reneighbor or exchange, compute, final kick --
    two library calls per step, no blocking check.  A drifting hot system reneighbors by itself several times; the
    trajectory is the one-GPU run's (whose own deferred flag fires on the same evidence)."""
    if style == "rebomos":
        s = S.jitter(S.replicate(S.rebomos_bulk_cell(), (2, 2, 1)), 0.05, seed=27)
        v0 = S.gaussian_velocities(s, 300.0, seed=5) + np.array([60.0, -30.0, 20.0])   # < 0.1 A per step: never late
        st = capi.STYLE_REBOMOS
    else:
        s = S.jitter(S.fcc_cell(4.045, 12, frac_type2=0.05, seed=19), 0.04, seed=20)
        s.mass[1:3] = capi.AeamFile(POT_AEAM).mass[:2]
        v0 = S.gaussian_velocities(s, 600.0, seed=6) + np.array([40.0, 25.0, -30.0])
        st = capi.STYLE_AEAM
    steps = 60

    def run(native):
        if st == capi.STYLE_REBOMOS:
            ctx, cutghost = _rebo_ctx()
            skin, map_ = 2.0, MAP
        else:
            ctx = capi.Context(0)
            af = capi.AeamFile(POT_AEAM)
            tabs = af.build()
            ctx.aeam_set_tables(tabs)
            ctx._af = (af, tabs)
            skin, map_ = 1.0, None
            cutghost = float(af.cut_table(tabs).max()) + skin
        tr = resident.NativeTransport(1, 0) if native else None
        d = resident.DeviceDomain(ctx, st, s, cutghost, skin, map_, v0=v0, transport=tr, self_remote=native)
        d.compute(0, 0)
        for step in range(1, steps + 1):
            last = step == steps
            d.step(1 if last else 0, 1 if last else 0, rebuild="halo" if native else "auto", defer_final=not last)
        th = d.thermo()
        tags, a = _by_tag(d, ("x", "v"))
        order = np.argsort(tags)
        info = ctx.dd_comm_step_info() if native else None
        builds, late = d.builds, d.dangerous
        ctx.close()
        return th, a["x"][order], a["v"][order], builds, late, info

    pth, px, pv, pb, pl, _ = run(False)
    nth, nx, nv, nb, nl, info = run(True)
    assert pb >= 3 and nb >= 3 and abs(nb - pb) <= 1          # both reneighbored by themselves, on the same evidence
    assert nl == 0 and pl == 0                                 # ... and never late
    assert info["reneighbors"] == nb
    if style == "aeam":
        assert info["aeam_phased"] > 0
    dx = nx - px
    dx -= np.round(s.box.x2lamda(dx + s.box.lo)) @ s.box.h.T
    assert np.abs(dx).max() < 1e-8
    assert np.abs(nv - pv).max() < 1e-7
    assert nth["pe"] == pytest.approx(pth["pe"], rel=1e-10)
    assert nth["ke"] == pytest.approx(pth["ke"], rel=1e-9)


@pytest.mark.parametrize("style", ["rebomos", "aeam"])
def test_pruned_rows_give_the_trajectory_of_the_rows_as_built(style, monkeypatch):
    """Dynamic pruning of the tile rows (tile_prune_kernel): a hot run with a narrow buffer prunes every few steps;
    positions and velocities after 60 steps must agree with the run that walks the rows as built to rounding (the
    dropped entries contribute exactly zero; the kept ones land on other lanes, so partial sums differ in the last
    bit), and no pruning may come late."""
    if style == "rebomos":
        s = S.replicate(S.rebomos_bulk_cell(), (3, 3, 1))
        temp, skin, map_ = 600.0, 2.0, MAP
    else:
        s = S.fcc_cell(4.045, 7, frac_type2=0.02, seed=3)
        temp, skin, map_ = 863.0, 1.0, None
    out = {}
    for tag, env in (("pruned", {"MDP_PRUNE": "1", "MDP_PRUNE_BUFFER": "0.25"}), ("adaptive", {"MDP_PRUNE": "1"}),
                     ("as built", {"MDP_PRUNE": "0"})):
        for k in ("MDP_PRUNE", "MDP_PRUNE_BUFFER"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        if style == "rebomos":
            ctx, cutghost = _rebo_ctx()
            st = capi.STYLE_REBOMOS
        else:
            af = capi.AeamFile(POT_AEAM)
            tabs = af.build()
            ctx = capi.Context(0)
            ctx.aeam_set_tables(tabs)
            s.mass[1:3] = af.mass
            cutghost = float(af.cut_table(tabs).max()) + 1.0
            st = capi.STYLE_AEAM
        v0 = S.gaussian_velocities(s, temp, seed=77)
        d = resident.DeviceDomain(ctx, st, s, cutghost, skin, map_, v0=v0)
        d.compute(0, 0)
        for step in range(60):
            d.step(0, 0, rebuild="auto")
        got = ctx.md_download(d.nlocal, want=("x", "v"))
        order = np.argsort(d.tags_local)
        stats = ctx.md_prune_stats()
        out[tag] = (got["x"][order], got["v"][order], stats, d.builds)
        ctx.close()
    xp, vp, sp, bp = out["pruned"]
    xb, vb, sb, bb = out["as built"]
    assert sp["prunings"] >= 4 and sp["late"] == 0 and sb["prunings"] == 0
    assert bp == bb
    assert np.abs(xp - xb).max() < 1e-9 and np.abs(vp - vb).max() < 1e-7
    xa_, va_, sa, ba = out["adaptive"]          # the buffer widens by itself when the trigger fires within a dozen computes
    assert sa["late"] == 0 and sa["buffer"] > 0.3 + 1e-9 and sa["prunings"] < sp["prunings"] and ba == bb
    assert np.abs(xa_ - xb).max() < 1e-9 and np.abs(va_ - vb).max() < 1e-7


def test_one_fast_atom_in_a_cold_crystal_keeps_the_pruned_rows_valid(monkeypatch):
    """The displacement votes behind the pruned rows and the style's own lists must count EVERY atom.  (Until round 6 the
    vote sat inside the branch only lane 0 of a wave takes, so one atom in 64 was looked at -- in a thermal system some
    lane-0 atom always moves about as far as the fastest, which hid it; profiles/prune_fuzz.py found a 5 000 K run where
    it did not.)  Here ONE S atom of a crystal at rest is shot at 40 A/ps through the lattice, and it is chosen so that
    it does not sit on lane 0 of its wave in the device's order: it crosses Lennard-Jones cutoffs of atoms that do not
    move at all.  The run on pruned rows must stay on the run that walks the rows as built."""
    s = S.replicate(S.rebomos_bulk_cell(), (3, 3, 2))
    # the device's order (a function of the positions): an S atom at a slot that is no multiple of 64, nor next to one
    ctx, cutghost = _rebo_ctx()
    d = resident.DeviceDomain(ctx, capi.STYLE_REBOMOS, s, cutghost, 2.0, MAP, v0=np.zeros((s.n, 3)))
    typ = ctx.md_download_int("type", d.nlocal)
    slot = next(k for k in range(100, d.nlocal) if typ[k] == 2 and 8 <= k % 64 <= 56)
    shot = int(d.tags_local[slot])
    ctx.close()
    out = {}
    for tag, env in (("pruned", {"MDP_PRUNE": "1", "MDP_PRUNE_BUFFER": "0.3"}), ("as built", {"MDP_PRUNE": "0"})):
        for k in ("MDP_PRUNE", "MDP_PRUNE_BUFFER"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx, cutghost = _rebo_ctx()
        v0 = np.zeros((s.n, 3))
        v0[np.nonzero(s.tag == shot)[0][0]] = np.array([25.0, 22.0, 22.0])          # 40 A/ps: 0.04 A per step
        d = resident.DeviceDomain(ctx, capi.STYLE_REBOMOS, s, cutghost, 2.0, MAP, v0=v0)
        tags = d.tags_local
        assert int(tags[slot]) == shot
        d.compute(0, 0)
        for step in range(70):
            d.step(0, 0, rebuild="auto")
        got = ctx.md_download(d.nlocal, want=("x", "v"))
        order = np.argsort(d.tags_local)
        out[tag] = (got["x"][order], got["v"][order], ctx.md_prune_stats(), int(tags[slot]))
        ctx.close()
    xp, vp, sp, tp = out["pruned"]
    xb, vb, sb, tb = out["as built"]
    assert tp == tb and sp["prunings"] >= 5 and sb["prunings"] == 0    # the projectile alone re-prunes the rows every few steps
    assert np.abs(xp - xb).max() < 1e-10 and np.abs(vp - vb).max() < 1e-8


def test_upload_x_invalidates_the_pruned_rows(monkeypatch):
    """mdp_md_upload_x rewrites the positions outside the integrator (here: every atom moved by up to 0.5 A, more
    than half of any pruning buffer).  The compute that follows must prune afresh and check the style's own lists
    before it walks them: forces and energy equal those of the run that walks the rows as built."""
    s = S.replicate(S.rebomos_bulk_cell(), (3, 3, 1))
    v0 = S.gaussian_velocities(s, 300.0, seed=5)
    jit = np.random.default_rng(12).uniform(-0.29, 0.29, size=(s.n, 3))
    out = {}
    for tag, env in (("pruned", {"MDP_PRUNE": "1", "MDP_PRUNE_BUFFER": "0.25"}), ("as built", {"MDP_PRUNE": "0"})):
        for k in ("MDP_PRUNE", "MDP_PRUNE_BUFFER"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx, cutghost = _rebo_ctx()
        d = resident.DeviceDomain(ctx, capi.STYLE_REBOMOS, s, cutghost, 2.0, MAP, v0=v0)
        d.compute(0, 0)
        for _ in range(6):
            d.step(0, 0, rebuild="auto")
        if tag == "pruned":
            assert ctx.md_prune_stats()["active"]
        tags = d.tags_local
        x = ctx.md_download(d.nlocal, want=("x",))["x"]
        ctx.md_upload_x(x + jit[tags - 1])
        d.compute(3, 1)
        got = ctx.md_download(d.nlocal, want=("f",))
        order = np.argsort(tags)
        out[tag] = (got["f"][order], d.thermo()["pe"], ctx.md_prune_stats())
        ctx.close()
    fp, ep, sp = out["pruned"]
    fb, eb, _ = out["as built"]
    assert sp["late"] == 0
    assert np.abs(fp - fb).max() < 1e-9
    assert ep == pytest.approx(eb, rel=1e-12)


def test_eight_bricks_hot_run_meets_the_oracle_at_the_seams(oracle):
    """12x12x12 replica (497 664 atoms) on 2x2x2 bricks at 300 K with a drift, 30 steps, a reneighboring (migration,
    re-derived ghosts, new lists) every 10: forces of ~500-atom blocks at the box corners, on the brick faces and
    corners and at random places equal the oracle's -- the decomposed, migrated, pruned state meets the oracle
    directly, not only the one-brick run."""
    import blockcheck
    P = oracle.rebomos_params(POT_REBOMOS)
    s = S.replicate(S.rebomos_bulk_cell(), (12, 12, 12))
    v0 = S.gaussian_velocities(s, 300.0, seed=23) + np.array([50.0, -35.0, 20.0])
    r = _run(8, s, v0, 30, 10)
    assert r["left"] > 200 and r["builds"] == 4
    pts = blockcheck.seeds(s.box, r["x"], n_random=1)
    worst, rows = blockcheck.check_blocks(s.box, r["x"], r["f"], s.type, s.tag, s.mass, pts,
                                          lambda cs: mdref.RebomosCPU(oracle, P, cs), n_interior=400, shell=11.0,
                                          margin=16.0, tol=1e-9)
    assert len(rows) >= 7


def test_slab_with_a_non_periodic_dimension(oracle):
    """`boundary p p f`: a 28 A thick MoS2 slab in a 56 A cell whose z direction is NOT periodic (mdp_dd_config
    .nonperiodic): no images across z, no wrap, the end bricks take whatever lies beyond the box.  One brick:
    forces equal the oracle's for the same slab (the oracle is periodic, the 28 A of vacuum keep the images out of
    range).  2x2x2 bricks (the z seam runs through the slab): same trajectory while atoms drift across the seam; one
    atom pushed out through the top of the box stays owned, unwrapped, by the upper bricks."""
    P = oracle.rebomos_params(POT_REBOMOS)
    b = S.replicate(S.rebomos_bulk_cell(), (4, 3, 2))
    box = S.Box(b.box.lo.copy(), b.box.prd * np.array([1.0, 1.0, 2.0]), b.box.tilt.copy())
    x = b.x + np.array([0.0, 0.0, 6.0])          # 6 A of vacuum below, 22 A above: periodic images 28 A apart
    s = S.jitter(S.System(box, x, b.type, b.tag, b.mass), 0.05, seed=31)
    nonper = (0, 0, 1)
    ctx, cutghost = _rebo_ctx()
    d = resident.DeviceDomain(ctx, capi.STYLE_REBOMOS, s, cutghost, 2.0, MAP, nonperiodic=nonper)
    d.compute(3, 1)
    th = d.thermo()
    tags, a = _by_tag(d, ("x", "f"))
    order = np.argsort(tags)
    o = mdref.RebomosCPU(oracle, P, S.System(box, S.wrap(box, s.x), s.type, s.tag, s.mass)).compute(S.wrap(box, s.x))
    assert np.abs(a["f"][order] - o["f_owned"]).max() < 1e-9
    assert th["pe"] == pytest.approx(o["eng"], rel=1e-10)
    # fewer ghosts than the fully periodic cell would have: none below and above the slab
    dper = resident.DeviceDomain(_rebo_ctx()[0], capi.STYLE_REBOMOS, s, cutghost, 2.0, MAP)
    assert d.nself < dper.nself
    dper.ctx.close()
    ctx.close()

    v0 = S.gaussian_velocities(s, 300.0, seed=4) + np.array([30.0, -20.0, 60.0])
    top = int(np.argmax(s.x[:, 2]))                             # an atom of the top layer evaporates: it leaves through
    v0[top] = np.array([0.0, 0.0, 800.0])                        # the top of the box within the run (24 A in 30 steps)

    def run(world):
        def rank_fn(r, make_tr):
            c, cg = _rebo_ctx()
            tr = make_tr(c) if world > 1 else None
            dd = resident.DeviceDomain(c, capi.STYLE_REBOMOS, s, cg, 2.0, MAP, v0=v0, transport=tr, nonperiodic=nonper)
            dd.compute(0, 0)
            left = 0
            for step in range(1, 31):
                rb = step % 3 == 0
                dd.step(0, 0, rebuild=rb)
                if rb:
                    left += c.dd_info()["left_last"]
            t, arr = _by_tag(dd, ("x", "v"))
            c.close()
            return t, arr["x"], arr["v"], left
        res = [rank_fn(0, None)] if world == 1 else resident.run_ranks(world, rank_fn)
        xx, vv = np.zeros((s.n, 3)), np.zeros((s.n, 3))
        seen = np.zeros(s.n, dtype=int)
        for t, xr, vr, _ in res:
            xx[t - 1], vv[t - 1] = xr, vr
            seen[t - 1] += 1
        assert np.all(seen == 1)
        return xx, vv, sum(r[3] for r in res)

    x1, v1, _ = run(1)
    x8, v8, left = run(8)
    assert left > 10                                           # atoms crossed the seams
    assert x1[top, 2] > box.lo[2] + box.prd[2]                 # the evaporated atom is above the box, not wrapped
    dx = x8 - x1
    lam = np.round(S.mul_upper(dx, box.hinv))
    lam[:, 2] = 0.0                                            # (no periodic image in z to forgive)
    dx -= S.mul_upper(lam, box.h)
    assert np.abs(dx).max() < 1e-8 and np.abs(v8 - v1).max() < 1e-7


def test_device_order_is_element_sorted_inside_every_run_of_32_atoms():
    """rebomos: after the device's reneighboring the owned atoms lie along the Hilbert curve of the brick, and inside
    every run of 32 consecutive atoms (one tile of the Lennard-Jones lists) the Mo atoms come first -- two-atom rows
    are then element-pure wherever the run allows, and the rows sharing a wave pad alike.  The runs themselves stay
    compact blobs of the curve."""
    s = S.jitter(S.replicate(S.rebomos_bulk_cell(), (3, 3, 2)), 0.05, seed=2)
    ctx, cutghost = _rebo_ctx()
    d = resident.DeviceDomain(ctx, capi.STYLE_REBOMOS, s, cutghost, 2.0, MAP)
    t = ctx.md_download_int("type", d.nlocal)
    x = ctx.md_download(d.nlocal, want=("x",))["x"]
    nrun = d.nlocal // 32
    for r in range(nrun):
        tt = t[32 * r:32 * r + 32]
        assert np.all(np.diff(tt) >= 0)                       # type 1 (Mo) first, then type 2 (S)
    ext = np.array([np.ptp(x[32 * r:32 * r + 32], axis=0).max() for r in range(nrun)])
    assert np.median(ext) < 0.45 * float(np.max(s.box.prd))   # compact runs (a random order spans the whole box)
    mixed = sum(1 for k in range(d.nlocal // 2) if t[2 * k] != t[2 * k + 1])
    assert mixed <= nrun                                      # at most one mixed two-atom row per run
    ctx.close()


def test_whole_step_calls_without_a_communicator_are_refused():
    """mdp_dd_comm_step_begin / _end need mdp_dd_comm_init (a one-GPU domain has no communicator): an error with its
    reason, nothing queued"""
    s = S.rebomos_bulk_cell()
    ctx, cutghost = _rebo_ctx()
    d = resident.DeviceDomain(ctx, capi.STYLE_REBOMOS, s, cutghost, 2.0, MAP)
    d.compute(0, 0)
    with pytest.raises(capi.MdpError, match="mdp_dd_comm_init not called"):
        ctx.dd_comm_step_begin(False, -1, 0, 0)
    with pytest.raises(capi.MdpError, match="mdp_dd_comm_init not called"):
        ctx.dd_comm_step_end(0, 0, False)
    d.step(0, 0)                                   # ... and the domain goes on as before
    assert np.isfinite(d.thermo()["pe"])
    ctx.close()

