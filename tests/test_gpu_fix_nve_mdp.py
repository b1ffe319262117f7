"""GPU: `fix nve/mdp` -- the plugins' second style (a fix registered from a plugin as USER-BFIELD/bfieldplugin.cpp:15-29
does), NVE integration on the device for the two pair styles: between two reneighborings of the host nothing per atom
crosses the link.  Through `plugin load` + `run` in the mini-host (the reference's log, agreement with the host's own
`fix nve` on hot runs with reneighborings, every neigh_modify setting) and through the C-ABI (mdp_hnve_*)."""
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, POT_REBOMOS
from lammps_plugins_amd.host import capi, system as S
from test_plugin_boundary import PKG, _run, _thermo_rows
import mdref
import oracle_bindings as ob

pytestmark = pytest.mark.gpu


def test_fix_nve_mdp_reproduces_the_reference_log():
    """in.rebomos-bulk with `fix integrate all nve/mdp`: log.rebomos-bulk.1:54-56"""
    log = json.load(open(os.path.join(GOLDEN, "rebomos_bulk_log.json")))
    rc, out, err = _run(script_file="examples/in.rebomos-bulk.nve-mdp.mi355x")
    assert rc == 0, err
    assert "Loaded 2 plugins from rebomosplugin.so" in out
    rows = _thermo_rows(out)
    assert [int(r[0]) for r in rows] == [0, 10, 20]
    for got, ref in zip(rows, log["thermo"]):
        assert got[1] == pytest.approx(ref["temp"], abs=1e-5 * max(1.0, abs(ref["temp"])) * 1e-2 + 6e-6)
        assert got[2] == pytest.approx(ref["press"], abs=6e-3)
        assert got[3] == pytest.approx(ref["pe"], abs=6e-5)
        assert got[4] == pytest.approx(ref["ke"], abs=6e-8)
    assert "Neighbor list builds = 0" in out


def _script(path, **subs):
    text = open(os.path.join(PKG, "examples", path)).read()
    for old, new in subs.items():
        assert old in text, old
        text = text.replace(old, new)
    return text


REBO_HOT = dict([("create_atoms 2 box basis 1 1 basis 2 1 basis 3 2 basis 4 2 basis 5 2 basis 6 2",
                  "create_atoms 2 box basis 1 1 basis 2 1 basis 3 2 basis 4 2 basis 5 2 basis 6 2\nreplicate 2 2 2"),
                 ("thermo 10", "velocity all create 1500.0 4928459\nneighbor 0.4 bin\nthermo 50"), ("run 20", "run 300")])


@pytest.mark.parametrize("neigh,fixargs", [("delay 1000000 every 1 check no", ""), ("every 1 delay 0 check yes", ""),
                                           ("every 5 delay 10 check yes", ""), ("every 1 delay 0 check yes", " hostcheck yes"),
                                           ("every 7 delay 0 check no", "")])
def test_rebomos_hot_run_equals_the_hosts_fix_nve(neigh, fixargs):
    """2 304 atoms from 1 500 K, 300 steps, 0.4 A of skin: the host reneighbors many times.  Under `check yes` the device's
    displacement check decides (the fix takes the host's own look at atom->x out of the steps and asks through
    force_reneighbor); `hostcheck yes` keeps the host looking, with x and v downloaded for it; `check no` rebuilds by the
    calendar and gets x and v for those steps.  The thermo rows are those of the host's own `fix nve` run to the printed digits."""
    base = _script("in.rebomos-bulk.mi355x", **REBO_HOT)
    rc0, out0, err0 = _run(base)
    assert rc0 == 0, err0
    dev = base.replace("fix integrate all nve", f"neigh_modify {neigh}\nfix integrate all nve/mdp{fixargs}")
    rc1, out1, err1 = _run(dev, env={"MDP_FIX_STATS": "1"})
    assert rc1 == 0, err1
    downloads = int(re.search(r"fix nve/mdp: (\d+) downloads", out1).group(1))
    took = "check yes decided on the device" in out1
    assert took == ("check yes" in neigh and not fixargs)
    if fixargs:
        assert downloads >= 300                               # the host looked at atom->x on every step
    elif "check yes" in neigh:
        assert 7 <= downloads < 60                            # output steps + the reneighborings the device asked for
    r0, r1 = _thermo_rows(out0), _thermo_rows(out1)
    assert [int(r[0]) for r in r1] == [0, 50, 100, 150, 200, 250, 300]
    builds = int(re.search(r"Neighbor list builds = (\d+)", out1).group(1))
    assert builds >= 2                                        # the run did reneighbor
    for a, b in zip(r0, r1):
        assert a[1] == pytest.approx(b[1], rel=2e-7)          # temp (8 digits printed; reneighborings fall on other steps)
        assert a[3] == pytest.approx(b[3], rel=2e-8)          # pe
        assert a[4] == pytest.approx(b[4], rel=2e-7)          # ke
        assert a[2] == pytest.approx(b[2], rel=1e-5, abs=0.5) # press


def test_aeam_run_equals_the_hosts_fix_nve():
    """sample.in's system (32 000 atoms, 863 K), 200 steps with reneighborings: `fix nve/mdp` against `fix nve`"""
    base = _script("in.aeam-alsi.mi355x", **{"run 400": "run 200"})
    rc0, out0, err0 = _run(base, timeout=600)
    assert rc0 == 0, err0
    rc1, out1, err1 = _run(script_file="examples/in.aeam-alsi.nve-mdp.mi355x".replace(".mi355x", ".mi355x"), timeout=600)
    assert rc1 == 0, err1
    r0, r1 = _thermo_rows(out0), _thermo_rows(out1)
    assert "Loaded 2 plugins from aeamplugin.so" in out1
    for a, b in zip(r0[:3], r1[:3]):                          # step temp etotal pe vol press at steps 0, 100, 200
        assert a[1] == pytest.approx(b[1], rel=1e-6)
        assert a[2] == pytest.approx(b[2], rel=1e-8)
        assert a[3] == pytest.approx(b[3], rel=1e-8)


def test_bricks_yes_on_one_rank_reproduces_the_reference_log():
    """`fix ID all nve/mdp bricks yes`: the library's decomposition with ONE brick runs the steps -- lists, reneighborings
    and the periodic images on the device, the host's Neighbor idle for the length of the run (log.rebomos-bulk.1:54-56)"""
    log = json.load(open(os.path.join(GOLDEN, "rebomos_bulk_log.json")))
    text = _script("in.rebomos-bulk.nve-mdp.mi355x", **{"fix integrate all nve/mdp": "fix integrate all nve/mdp bricks yes"})
    rc, out, err = _run(text, env={"MDP_FIX_STATS": "1"})
    assert rc == 0, err
    rows = _thermo_rows(out)
    assert [int(r[0]) for r in rows] == [0, 10, 20]
    for got, ref in zip(rows, log["thermo"]):
        assert got[1] == pytest.approx(ref["temp"], abs=6e-6)
        assert got[2] == pytest.approx(ref["press"], abs=6e-3)
        assert got[3] == pytest.approx(ref["pe"], abs=6e-5)
        assert got[4] == pytest.approx(ref["ke"], abs=6e-8)
    assert "fix nve/mdp: 1 bricks, 0 reneighborings on the device, 2 returns of the atoms to the host" in out


def test_bricks_yes_hot_runs_equal_the_hosts_fix_nve_and_reneighbor_on_the_device():
    """the hot MoS2 cell (300 steps, 0.4 A of skin) and sample.in's alloy (200 steps from 863 K) with `bricks yes` on one
    rank: thermo rows of the host's own `fix nve`; every reneighboring happened on the device, the host built its lists
    once (at setup)"""
    base = _script("in.rebomos-bulk.mi355x", **REBO_HOT)
    rc0, out0, err0 = _run(base)
    assert rc0 == 0, err0
    rc1, out1, err1 = _run(base.replace("fix integrate all nve", "fix integrate all nve/mdp bricks yes"), env={"MDP_FIX_STATS": "1"})
    assert rc1 == 0, err1
    r0, r1 = _thermo_rows(out0), _thermo_rows(out1)
    assert len(r0) == len(r1) == 7
    for a, b in zip(r1, r0):
        for u, v in zip(a, b):
            assert u == pytest.approx(v, rel=5e-7, abs=1e-5)
    m = re.search(r"fix nve/mdp: 1 bricks, (\d+) reneighborings on the device, (\d+) returns", out1)
    assert m and int(m.group(1)) > 5 and int(m.group(2)) == 6     # (output steps 50 ... 300)
    assert "Neighbor list builds = 0" in out1                 # (the host's, during the run)
    a0 = _script("in.aeam-alsi.mi355x", **{"run 400": "run 200"})
    rc2, out2, err2 = _run(a0, timeout=600)
    assert rc2 == 0, err2
    rc3, out3, err3 = _run(_script("in.aeam-alsi.nve-mdp.mi355x", **{"run 400": "run 200", "fix integrate all nve/mdp":
                                                                       "fix integrate all nve/mdp bricks yes"}), timeout=600)
    assert rc3 == 0, err3
    for a, b in zip(_thermo_rows(out2)[:3], _thermo_rows(out3)[:3]):
        assert a[1] == pytest.approx(b[1], rel=1e-6) and a[2] == pytest.approx(b[2], rel=1e-8) and a[3] == pytest.approx(b[3], rel=1e-8)


def test_fix_nve_mdp_keywords_are_checked():
    for bad in ("bricks maybe", "hostcheck", "speed fast"):
        rc, out, err = _run(_script("in.rebomos-bulk.nve-mdp.mi355x", **{"fix integrate all nve/mdp": "fix integrate all nve/mdp " + bad}))
        assert rc == 1 and "Illegal fix nve/mdp command" in err
    rc, out, err = _run(_script("in.rebomos-bulk.nve-mdp.mi355x", **{"fix integrate all nve/mdp": "fix integrate all nve/mdp bricks yes hostcheck yes"}))
    assert rc == 1 and "hostcheck yes needs the host's arrays current" in err


def test_fix_nve_mdp_needs_a_pair_style_of_the_plugin():
    script = _script("in.rebomos-bulk.nve-mdp.mi355x").replace("pair_style rebomos\n", "").replace(
        "pair_coeff * * ../tests/golden/potentials/MoS.REBO.set5b M S\n", "")
    rc, out, err = _run(script)
    assert rc == 1 and ("no pair style" in err or "requires a pair style" in err)


def test_hnve_calls_follow_a_host_loop_around_the_oracle(oracle):
    """the C-ABI underneath: mdp_hnve_setup / _upload_v / _initial / compute with f == NULL / _final / _download against
    velocity-Verlet on the host around the oracle's compute(), 60 steps from 600 K (positions 1e-9 A)"""
    P = oracle.rebomos_params(POT_REBOMOS)
    s = S.replicate(S.rebomos_bulk_cell(), (2, 1, 1))
    v0 = S.gaussian_velocities(s, 600.0, seed=77)
    eng = mdref.RebomosCPU(oracle, P, s, skin=2.0)
    _, xref, vref = mdref.nve(eng, s, 60, v0=v0)
    c = capi.Context(0)
    c.rebomos_set_params(ob.product_rebomos_params(P))
    c.set_box_host(s.box)
    c.set_atoms_host(eng.nlocal, eng.all_positions(s.x), eng.type_all, eng.tag_all, 2, map_=[0, 0, 1])
    assert c.host_ghosts_derived()
    c.set_skin(2.0)
    c.hnve_setup(0.001, S.FTM2V, s.mass)
    c.hnve_upload_v(v0)
    c.rebomos_compute_host(eng.nlocal, eflag=0, vflag=0)      # forces of the initial state (stay on the device too)
    moved_any = False
    for _ in range(60):
        moved, late = c.hnve_initial()
        moved_any |= moved
        assert not late
        c._ck(c.L.mdp_rebomos_compute_host(c.h, 0, 0, None, None, None, None, None))
        c.hnve_final()
    got = c.hnve_download(eng.nlocal)
    assert np.abs(got["x"] - xref).max() < 1e-9
    assert np.abs(got["v"] - vref).max() < 1e-8
    assert np.abs(got["f"] - eng.compute(xref)["f_owned"]).max() < 1e-8
    assert not moved_any                                       # 60 steps at 600 K stay inside half the 2 A skin
    c.close()
