"""GPU parity: REBO-MoS with its lists taken from the HOST's neighbor list (mdp_rebomos_host_list /
MDP_REBOMOS_HOST_LIST=1).  The reference walks the rows LAMMPS built (REBO_neigh, pair_rebomos.cpp:281-352 with
:328-330; FLJ, :490-495), so pairs the host left out -- `neigh_modify exclude`, special bonds -- are no pairs of the
style.  The oracle takes the list as an argument; here both sides get the SAME list with a group's internal pairs
removed.  Tolerances as tests/test_gpu_rebomos.py."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, POT_REBOMOS, ROOT
from lammps_plugins_amd.host import capi, system as S
import mdref
import oracle_bindings

pytestmark = pytest.mark.gpu
F_TOL, E_TOL = 1e-9, 1e-9
PKG = os.path.join(ROOT, "lammps-plugins_amd")


@pytest.fixture(scope="module")
def P(oracle):
    return oracle.rebomos_params(POT_REBOMOS)


@pytest.fixture()
def ctx(P):
    c = capi.Context(0)
    c.rebomos_set_params(oracle_bindings.product_rebomos_params(P))
    yield c
    c.close()


def _without_pairs(eng, drop):
    """the engine's CSR list without the entries (i, j) for which drop(i, j-array) is true (both directions are rows
    of their own in a full list: drop must be symmetric)"""
    nn, off, nb = eng.nn, eng.off, eng.nb
    rows = []
    for i in range(len(nn)):
        r = nb[off[i]:off[i] + nn[i]]
        rows.append(r[~drop(i, r)])
    eng.nn = np.array([len(r) for r in rows], dtype=np.int32)
    eng.off = np.zeros(len(rows) + 1, dtype=np.int64)
    eng.off[1:] = np.cumsum(eng.nn)
    eng.nb = np.concatenate(rows).astype(np.int32) if rows else np.zeros(0, dtype=np.int32)


def _gpu(ctx, eng, x, first=True, eflag=3, vflag=1, paged=False):
    xa = eng.all_positions(x)
    if first:
        ctx.set_atoms_host(eng.nlocal, xa, eng.type_all, eng.tag_all, 2, map_=[0, 0, 1])
        if paged:   # the LAMMPS shape: ilist / numneigh / firstneigh, owned rows then ghost rows
            rows = [eng.nb[eng.off[i]:eng.off[i] + eng.nn[i]] for i in range(len(eng.nn))]
            gnum = len(eng.nn) - eng.nlocal
            ctx.set_neighbors_paged_host(eng.nlocal, gnum, np.arange(len(eng.nn)), eng.nn, rows, 2.0)
        else:
            ctx.set_neighbors_csr_host(eng.nn, eng.off, eng.nb, 2.0)
    else:
        ctx.set_positions_host(xa)
    return ctx.rebomos_compute_host(eng.nlocal, eflag=eflag, vflag=vflag)


def _compare(g, o):
    assert np.abs(g["f"] - o["f_owned"]).max() < F_TOL
    assert g["eng"] == pytest.approx(o["eng"], rel=1e-10)
    assert np.abs(g["eatom"] - o["eatom_owned"]).max() < E_TOL
    assert np.allclose(g["virial"], o["virial_fdotr"], rtol=1e-9, atol=1e-7)


def _system():
    return S.jitter(S.scale(S.replicate(S.rebomos_bulk_cell(), (2, 1, 1)), 1.06), 0.12, seed=31)


@pytest.mark.parametrize("paged", [False, True])
def test_lists_from_the_plain_host_list_give_the_default_result(ctx, oracle, P, paged):
    s = _system()
    eng = mdref.RebomosCPU(oracle, P, s)
    ctx.rebomos_host_list(True)
    g = _gpu(ctx, eng, s.x, paged=paged)
    _compare(g, eng.compute(s.x))
    assert ctx.rebomos_list_info()["tiled"]


def test_excluded_group_against_the_oracle_fed_the_same_list(ctx, oracle, P):
    """`neigh_modify exclude group G G` with G = a slab of the cell (atoms by tag, their periodic images included): no
    REBO bonds, no bond-order neighbours and no LJ pairs inside G -- in the oracle because the list lacks them, on the
    device because its lists are subsets of the same list.  Then the atoms move (no new list from the host) far enough
    for the style to rebuild its trimmed lists from the rows it was given."""
    s = _system()
    eng = mdref.RebomosCPU(oracle, P, s)
    full = eng.compute(s.x)
    in_g = s.x[:, 0] < s.box.lo[0] + 0.4 * s.box.prd[0]            # by owned atom; images inherit through the tag
    g_tag = np.zeros(int(eng.tag_all.max()) + 1, dtype=bool)
    g_tag[s.tag[in_g]] = True
    member = g_tag[eng.tag_all]
    assert 0.2 * s.n < in_g.sum() < 0.6 * s.n
    _without_pairs(eng, lambda i, js: member[i] & member[js])
    ctx.rebomos_host_list(True)
    g = _gpu(ctx, eng, s.x)
    o = eng.compute(s.x)
    _compare(g, o)
    assert abs(o["eng"] - full["eng"]) > 1.0                       # the exclusion is no small thing
    assert np.abs(o["f_owned"] - full["f_owned"]).max() > 0.1
    # new positions, same list: 0.6 A of shear across the cell, beyond half the style's inner skin (1.0 A at the start)
    x2 = s.x.copy()
    x2[:, 1] += 0.6 * np.sin(2 * np.pi * (s.x[:, 0] - s.box.lo[0]) / s.box.prd[0])
    builds0 = ctx.rebomos_list_info()["builds"]
    g2 = _gpu(ctx, eng, x2, first=False)
    _compare(g2, eng.compute(x2))
    assert ctx.rebomos_list_info()["builds"] > builds0


def test_default_path_refuses_the_same_list(ctx, oracle, P):
    s = _system()
    eng = mdref.RebomosCPU(oracle, P, s)
    member = (eng.tag_all % 3) == 0
    _without_pairs(eng, lambda i, js: member[i] & member[js])
    xa = eng.all_positions(s.x)
    ctx.rebomos_host_list(False)
    ctx.set_atoms_host(eng.nlocal, xa, eng.type_all, eng.tag_all, 2, map_=[0, 0, 1])
    ctx.set_skin(2.0)
    rows = [eng.nb[eng.off[i]:eng.off[i] + eng.nn[i]] for i in range(len(eng.nn))]
    with pytest.raises(capi.MdpError, match="not the plain geometric list"):
        ctx.rebomos_check_host_list(np.arange(eng.nlocal), eng.nn, rows, P.cut3rebo + 2.0)


def test_host_list_mode_without_a_list_is_an_error(ctx, oracle, P):
    s = _system()
    eng = mdref.RebomosCPU(oracle, P, s)
    ctx.rebomos_host_list(True)
    ctx.set_atoms_host(eng.nlocal, eng.all_positions(s.x), eng.type_all, eng.tag_all, 2, map_=[0, 0, 1])
    ctx.set_skin(2.0)
    with pytest.raises(capi.MdpError, match="no list was handed over"):
        ctx.rebomos_compute_host(eng.nlocal)


def _script():
    """examples/in.rebomos-bulk.mi355x (the system of USER-REBOMOS/in.rebomos-bulk) with Mo-Mo pairs excluded"""
    text = open(os.path.join(PKG, "examples", "in.rebomos-bulk.mi355x")).read()
    assert "\nrun 20" in text
    return text.replace("\nrun 20", "\nneigh_modify exclude type 1 1\nrun 20")


def _minilmp(env_extra):
    env = dict(os.environ, **env_extra)
    p = subprocess.run([os.path.join(PKG, "minilmp")], input=_script(), capture_output=True, text=True, cwd=PKG, env=env,
                       timeout=300)
    return p.returncode, p.stdout, p.stderr


def test_plugin_with_neigh_modify_exclude(oracle, P):
    """`neigh_modify exclude type 1 1` (no Mo-Mo entries in the host's list) through `plugin load`: the default path
    stops the run, MDP_REBOMOS_HOST_LIST=1 runs it and prints the oracle's energy for the same list at step 0"""
    rc, out, err = _minilmp({})
    assert rc != 0 and "not the plain geometric list" in err
    rc, out, err = _minilmp({"MDP_REBOMOS_HOST_LIST": "1"})
    assert rc == 0, err
    rows = [[float(v) for v in l.split()] for l in out.splitlines() if re.fullmatch(r"\s*\d+(\s+[-+0-9.eE]+){6}\s*", l)]
    assert [int(r[0]) for r in rows] == [0, 10, 20]
    s = S.rebomos_bulk_cell()
    eng = mdref.RebomosCPU(oracle, P, s)
    plain = eng.compute(s.x)["eng"]
    mo = eng.type_all == 1
    _without_pairs(eng, lambda i, js: mo[i] & mo[js])
    o = eng.compute(s.x)
    assert abs(o["eng"] - plain) > 1.0
    assert rows[0][3] == pytest.approx(o["eng"], abs=6e-5)          # (8 significant digits are printed)
    # ... and the oracle's NVE trajectory with that list (the layers fly apart without their Mo-Mo terms: 20 steps
    # move no atom by half the skin, so the list of step 0 stays the host's list)
    ref, _, _ = mdref.nve(eng, s, 20, thermo_every=10)
    for got, want in zip(rows, ref):
        assert got[3] == pytest.approx(want["pe"], abs=6e-5)
        assert got[4] == pytest.approx(want["ke"], abs=6e-6)
    assert "Neighbor list builds = 0" in out


def test_fix_nve_mdp_is_refused_with_the_host_list():
    """the integrator on the device keeps atoms and images there; lists from the host's rows need the host's atoms"""
    text = open(os.path.join(PKG, "examples", "in.rebomos-bulk.nve-mdp.mi355x")).read()
    env = dict(os.environ, MDP_REBOMOS_HOST_LIST="1")
    p = subprocess.run([os.path.join(PKG, "minilmp")], input=text, capture_output=True, text=True, cwd=PKG, env=env, timeout=300)
    assert p.returncode != 0
    assert "cannot be combined with MDP_REBOMOS_HOST_LIST=1" in p.stderr

