"""GPU parity on edge cases (SURVEY.md Appendix C "A-edge" and REBO analogues): empty domains,
isolated atoms, dimers/trimers, pairs sitting just inside / outside each cutoff, clamp branches.
Small hand-built clusters in a large periodic box, host mode, against the CPU oracle."""
import numpy as np
import pytest

from conftest import POT_AEAM, POT_REBOMOS
from lammps_plugins_amd.host import capi, system as S
import mdref
import oracle_bindings as ob

pytestmark = pytest.mark.gpu

BOX = S.Box(np.zeros(3), np.array([60.0, 60.0, 60.0]), np.zeros(3))


def _sys(x, types, mass=(0.0, 95.95, 32.065)):
    x = np.asarray(x, dtype=float).reshape(-1, 3) + 30.0
    t = np.asarray(types, dtype=np.int32)
    return S.System(BOX, x, t, np.arange(1, len(x) + 1, dtype=np.int32), np.array(mass))


@pytest.fixture(scope="module")
def rebo(oracle):
    P = oracle.rebomos_params(POT_REBOMOS)
    ctx = capi.Context(0)
    ctx.rebomos_set_params(capi.read_rebomos_file(POT_REBOMOS))
    yield oracle, P, ctx
    ctx.close()


@pytest.fixture(scope="module")
def aeam(oracle):
    T = oracle.aeam_pot(POT_AEAM)
    af = capi.AeamFile(POT_AEAM)
    tabs = af.build()
    ctx = capi.Context(0)
    ctx.aeam_set_tables(tabs)
    yield oracle, T, ctx, (af, tabs)
    ctx.close()


def _rebo_both(rebo, s):
    oracle, P, ctx = rebo
    eng = mdref.RebomosCPU(oracle, P, s)
    xa = eng.all_positions(s.x)
    ctx.set_atoms_host(eng.nlocal, xa, eng.type_all, eng.tag_all, 2, map_=[0, 0, 1])
    ctx.set_skin(2.0)
    return ctx.rebomos_compute_host(eng.nlocal), eng.compute(s.x)


REBO_CASES = {
    "single_Mo": ([[0, 0, 0]], [1]),
    "two_far": ([[0, 0, 0], [20, 0, 0]], [1, 2]),
    "MoS_dimer": ([[0, 0, 0], [2.41, 0, 0]], [1, 2]),
    "MoMo_in_switch": ([[0, 0, 0], [3.65, 0, 0]], [1, 1]),          # rcmin 3.5 < r < rcmax 3.8
    "SS_in_switch": ([[0, 0, 0], [2.6, 0.3, 0]], [2, 2]),           # 2.3 < r < 3.0
    "MoS_just_inside_rcmax": ([[0, 0, 0], [3.05 - 1e-7, 0, 0]], [1, 2]),   # w ~ 3e-15 <= TOL: bond skipped
    "MoS_just_outside_rcmax": ([[0, 0, 0], [3.05 + 1e-7, 0, 0]], [1, 2]),
    "LJ_cubic_branch": ([[0, 0, 0], [3.7, 0, 0]], [2, 1]),          # rcLJmin 2.75 <= r < 0.95 sigma_MS = 3.48?  no: LJ 12-6
    "LJ_cubic_MoMo": ([[0, 0, 0], [3.9, 0, 0]], [1, 1]),            # 3.5 <= r < 0.95*4.2 = 3.99 : cubic
    "LJ_at_rcLJmax": ([[0, 0, 0], [10.5, 0, 0], [0, 10.5 + 1e-9, 0]], [1, 1, 1]),   # one inside, one outside 2.5 sigma
    "S_Mo_S_linear": ([[0, 0, 0], [2.4, 0, 0], [-2.4, 0, 0]], [1, 2, 2]),      # cos = -1 clamp
    "S_Mo_S_narrow": ([[0, 0, 0], [2.4, 0, 0], [2.2, 0.9, 0]], [1, 2, 2]),     # cos > 0.5: blended G
    "MoS2_unit": ([[0, 0, 0], [1.84, 0, 1.58], [1.84, 0, -1.58], [-0.92, 1.59, 1.58], [-0.92, 1.59, -1.58],
                   [-0.92, -1.59, 1.58], [-0.92, -1.59, -1.58]], [1, 2, 2, 2, 2, 2, 2]),
}


@pytest.mark.parametrize("name", sorted(REBO_CASES))
def test_rebomos_small_clusters(rebo, name):
    x, t = REBO_CASES[name]
    g, o = _rebo_both(rebo, _sys(x, t))
    assert np.abs(g["f"] - o["f_owned"]).max() < 1e-9
    assert abs(g["eng"] - o["eng"]) < 1e-10 * max(1.0, abs(o["eng"]))
    assert np.abs(g["eatom"] - o["eatom_owned"]).max() < 1e-10
    assert np.allclose(g["virial"], o["virial_fdotr"], rtol=1e-9, atol=1e-9)


def test_rebomos_empty_domain(rebo):
    oracle, P, ctx = rebo
    ctx.set_atoms_host(0, np.zeros((0, 3)), np.zeros(0, np.int32), np.zeros(0, np.int32), 2, map_=[0, 0, 1])
    ctx.set_skin(2.0)
    g = ctx.rebomos_compute_host(0)
    assert g["eng"] == 0.0 and g["f"].shape == (0, 3)


def _aeam_both(aeam, s):
    oracle, T, ctx, _ = aeam
    eng = mdref.AeamCPU(oracle, T, s)
    xa = eng.all_positions(s.x)
    nall, nloc = len(xa), eng.nlocal
    ctx.set_atoms_host(nloc, xa, eng.type_all, eng.tag_all, 2, map_=None)
    ctx.set_neighbors_csr_host(eng.nn, eng.off, eng.nb, 1.0)
    d = ctx.aeam_density_host(nloc, eflag=3)
    fp_all = np.concatenate([d["fp"], d["fp"][eng.owner]])
    r = ctx.aeam_force_host(nall, nloc, fp_all, eflag=3, vflag=1)
    g = dict(f=ob.fold_ghost_forces(r["f"], eng.owner, nloc), eng=d["eng"] + r["eng"], virial=r["virial"],
             eatom=d["eatom"] + r["eatom"], rho=d["rho"])
    return g, eng.compute(s.x)


AEAM_CASES = {
    "isolated_Si": ([[0, 0, 0]], [2]),                               # rho = 0 -> Fptmp = 0, row clamp m = 1
    "isolated_Al": ([[0, 0, 0]], [1]),
    "Si_one_neighbour": ([[0, 0, 0], [2.6, 0, 0]], [2, 1]),          # no triplet
    "SiSi_dimer": ([[0, 0, 0], [2.35, 0, 0]], [2, 2]),
    "AlAl_below_cut": ([[0, 0, 0], [6.5 - 1e-9, 0, 0]], [1, 1]),     # last table row, p clamp
    "AlAl_above_cut": ([[0, 0, 0], [6.5 + 1e-9, 0, 0]], [1, 1]),
    "AlSi_at_cut": ([[0, 0, 0], [4.18 - 1e-9, 0, 0], [0, 4.18 + 1e-9, 0]], [2, 1, 1]),
    "SiSi_between_cutdec_and_cut": ([[0, 0, 0], [4.5, 0, 0], [0, 2.5, 0]], [2, 2, 1]),   # 5.28-1.5 < r < 5.28
    "Si_three_Al": ([[0, 0, 0], [2.5, 0, 0], [-1.2, 2.2, 0], [-1.2, -2.2, 0.3]], [2, 1, 1, 1]),
    "Si_mixed_neighbours": ([[0, 0, 0], [2.4, 0, 0], [-1.2, 2.1, 0], [0, -1, 2.2], [0.5, 0.5, -2.4]], [2, 2, 1, 2, 1]),
    "Al_close_pair": ([[0, 0, 0], [1.9, 0, 0]], [1, 1]),              # large density: embedding table upper rows
}


@pytest.mark.parametrize("name", sorted(AEAM_CASES))
def test_aeam_small_clusters(aeam, name):
    x, t = AEAM_CASES[name]
    g, o = _aeam_both(aeam, _sys(x, t, mass=(0.0, 27.0, 28.0)))
    scale = max(1.0, np.abs(o["f_owned"]).max())
    assert np.abs(g["rho"] - o["rho"][:len(t)]).max() < 1e-11 * max(1.0, np.abs(o["rho"]).max())
    assert np.abs(g["f"] - o["f_owned"]).max() < 1e-9 * scale
    assert abs(g["eng"] - o["eng"]) < 1e-10 * max(1.0, abs(o["eng"]))
    assert np.abs(g["eatom"] - o["eatom"][:len(t)]).max() < 1e-10 * max(1.0, abs(o["eng"]))
    assert np.allclose(g["virial"], o["virial_fdotr"], rtol=1e-9, atol=1e-9 * scale)


def test_rebomos_sparse_gas_in_a_large_box(rebo):
    """64 atoms scattered over a 200 A box, some of them in bonded pairs and trimers: the one or two tiles they form
    span the whole cell grid (about 30 x 30 rows of cells), so the tile-list builder goes through several blocks of
    its row table and most candidate rows are empty."""
    rng = np.random.default_rng(4242)
    x = rng.random((40, 3)) * 180.0 - 90.0
    extra, types = [], [int(t) for t in rng.integers(1, 3, size=40)]
    for k in range(8):                                   # Mo-S dimers and S-Mo-S trimers next to some of the atoms
        extra.append(x[k] + [2.41, 0.0, 0.0])
        types[k] = 1
        types.append(2)
    for k in range(8, 16):
        extra.append(x[k] + [0.0, 2.35, 0.4])
        extra.append(x[k] + [0.3, -2.3, 0.5])
        types[k] = 1
        types += [2, 2]
    x = np.vstack([x, np.array(extra)])
    box = S.Box(np.zeros(3), np.array([200.0, 200.0, 200.0]), np.zeros(3))
    s = S.System(box, x + 100.0, np.asarray(types, dtype=np.int32), np.arange(1, len(x) + 1, dtype=np.int32),
                 np.array((0.0, 95.95, 32.065)))
    g, o = _rebo_both(rebo, s)
    assert np.abs(o["f_owned"]).max() > 0.1              # the bonded atoms do feel forces
    assert np.abs(g["f"] - o["f_owned"]).max() < 1e-9
    assert abs(g["eng"] - o["eng"]) < 1e-10 * max(1.0, abs(o["eng"]))
    assert np.abs(g["eatom"] - o["eatom_owned"]).max() < 1e-10
