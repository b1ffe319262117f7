"""GPU: the plugins in HOST MODE on several ranks -- `minilmp -np N`, N ranks as N threads with their own host objects, a
brick of the box each on LAMMPS' processor grid, ghosts from the neighbouring bricks, migration at reneighborings, and the
pair style's pack / unpack callbacks moving fp BETWEEN ranks (pair_aeam.cpp:946-990 as Comm::forward_comm drives them).
Every rank has its own device context on the one card.  Known answers: log.rebomos-bulk.4:22 (2 by 2 by 1 grid), :54-56
(the thermo rows), :72-79 (Nlocal 72 each, Nghost 2771.5 ave / 2775 max / 2768 min, FullNghs 35712 each)."""
import json
import os
import re

import pytest

from conftest import GOLDEN
from test_plugin_boundary import PKG, _run, _thermo_rows

pytestmark = pytest.mark.gpu


def test_four_ranks_reproduce_the_reference_4_rank_log():
    log = json.load(open(os.path.join(GOLDEN, "rebomos_bulk_log.json")))
    rc, out, err = _run(script_file="examples/in.rebomos-bulk.mi355x", np=4)
    assert rc == 0, err
    assert "  2 by 2 by 1 MPI processor grid" in out                       # log.rebomos-bulk.4:22
    assert out.count("Loaded 2 plugins from rebomosplugin.so") == 1         # (ranks other than 0 keep quiet)
    rows = _thermo_rows(out)
    assert [int(r[0]) for r in rows] == [0, 10, 20]
    for got, ref in zip(rows, log["thermo"]):                               # log.rebomos-bulk.4:54-56 = .1:54-56
        assert got[1] == pytest.approx(ref["temp"], abs=6e-6)
        assert got[2] == pytest.approx(ref["press"], abs=6e-3)
        assert got[3] == pytest.approx(ref["pe"], abs=6e-5)
        assert got[4] == pytest.approx(ref["ke"], abs=6e-8)
    assert "Loop time of" in out and "on 4 procs for 20 steps with 288 atoms" in out
    assert re.search(r"Nlocal:\s+72 ave\s+72 max\s+72 min", out)            # log.rebomos-bulk.4:72
    assert re.search(r"Nghost:\s+2771.5 ave\s+2775 max\s+2768 min", out)    # :74
    assert re.search(r"FullNghs:\s+35712 ave\s+35712 max\s+35712 min", out)  # :78
    assert "Total # of neighbors = 142848" in out and "Ave neighs/atom = 496" in out


def _aeam(steps=100, thermo=25):
    text = open(os.path.join(PKG, "examples", "in.aeam-alsi.mi355x")).read()
    assert "run 400" in text and "thermo 100" in text
    return text.replace("run 400", "run %d" % steps).replace("thermo 100", "thermo %d" % thermo)


@pytest.mark.parametrize("np", [2, 4, 8])
def test_aeam_sample_system_on_n_ranks_prints_the_one_rank_thermo(np):
    """sample.in's system (32 000 atoms, 0.75 % Si, 863 K): fp travels between ranks through the style's callbacks, the
    three-body forces on ghosts come back through the host's reverse communication, atoms change ranks at the three
    reneighborings.  The thermo rows are the one-rank run's to the printed digits."""
    rc1, out1, err1 = _run(_aeam())
    assert rc1 == 0, err1
    rc, out, err = _run(_aeam(), np=np)
    assert rc == 0, err
    r1, rn = _thermo_rows(out1), _thermo_rows(out)
    assert len(r1) == len(rn) == 5
    for a, b in zip(rn, r1):
        assert a[0] == b[0]
        for u, v in zip(a[1:], b[1:]):
            assert u == pytest.approx(v, rel=2e-8, abs=1e-6)
    assert "Neighbor list builds = 3" in out and "Neighbor list builds = 3" in out1
    counts = [int(x) for x in re.findall(r"rank \d+: Nlocal (\d+)", out)]
    assert len(counts) == np and sum(counts) == 32000 and len(set(counts)) > 1   # (atoms migrated: the bricks differ)
    assert "Total # of neighbors = %s" % re.search(r"FullNghs:\s+(\d+)", out1).group(1) in out


def test_an_error_on_every_rank_is_said_once_and_stops_all_ranks():
    text = open(os.path.join(PKG, "examples", "in.rebomos-bulk.mi355x")).read().replace("pair_coeff * * ../tests/golden/potentials/MoS.REBO.set5b M S",
                                                                                       "pair_coeff * * ../tests/golden/potentials/MoS.REBO.set5b M X")
    rc, out, err = _run(text, np=4, timeout=60)
    assert rc == 1
    assert err.count("Incorrect args for pair coefficients") == 1


# ---- fix nve/mdp on several ranks: the library's bricks behind the plugin surface -------------------------------------
# (the ranks share the one card, so RCCL is the test double of tests/native, as in test_gpu_native_ranks.py)

def _double_env():
    import subprocess
    from lammps_plugins_amd.host import capi
    if not os.path.exists(capi.FAKE_RCCL):
        subprocess.run(["make", "-C", PKG, "rccl-double"], check=True)
    return dict(MDP_RCCL_LIBRARY=capi.FAKE_RCCL, MDP_FAKE_RCCL_TIMEOUT_S="60", MDP_FIX_STATS="1")


def test_fix_nve_mdp_on_four_ranks_reproduces_the_reference_4_rank_log():
    """in.rebomos-bulk with `fix integrate all nve/mdp` on 2 x 2 x 1 ranks: each rank's 72 atoms go to its brick on the
    device at setup, the steps run there (mdp_dd_comm_step_begin in initial_integrate, _end in Pair::compute), the thermo
    rows are log.rebomos-bulk.4:54-56"""
    log = json.load(open(os.path.join(GOLDEN, "rebomos_bulk_log.json")))
    rc, out, err = _run(script_file="examples/in.rebomos-bulk.nve-mdp.mi355x", np=4, env=_double_env())
    assert rc == 0, err
    assert "  2 by 2 by 1 MPI processor grid" in out
    rows = _thermo_rows(out)
    assert [int(r[0]) for r in rows] == [0, 10, 20]
    for got, ref in zip(rows, log["thermo"]):
        assert got[1] == pytest.approx(ref["temp"], abs=6e-6)
        assert got[2] == pytest.approx(ref["press"], abs=6e-3)
        assert got[3] == pytest.approx(ref["pe"], abs=6e-5)
        assert got[4] == pytest.approx(ref["ke"], abs=6e-8)
    assert re.search(r"fix nve/mdp: 4 bricks, \d+ reneighborings on the device, 2 returns of the atoms to the host", out)
    assert "Neighbor list builds = 0" in out                               # (the host built its lists at setup only)


def _aeam_mdp(tail=None):
    text = open(os.path.join(PKG, "examples", "in.aeam-alsi.nve-mdp.mi355x")).read()
    assert "run 400" in text and "thermo 100" in text
    return text.replace("thermo 100", "thermo 25").replace("run 400", tail or "run 100")


@pytest.mark.parametrize("np", [2, 4, 8])
def test_fix_nve_mdp_aeam_on_n_ranks_prints_the_one_rank_thermo(np):
    """sample.in's system with its neighbor settings, 100 steps from 863 K: the bricks reneighbor by the flag that rides
    in the halo and atoms change bricks; on output steps the host gets back the atoms each brick owns by then (other
    counts than it handed over) and prints the thermo rows of the one-rank run"""
    rc1, out1, err1 = _run(_aeam_mdp())
    assert rc1 == 0, err1
    rc, out, err = _run(_aeam_mdp(), np=np, env=_double_env())
    assert rc == 0, err
    r1, rn = _thermo_rows(out1), _thermo_rows(out)
    assert len(r1) == len(rn) == 5
    for a, b in zip(rn, r1):
        assert a[0] == b[0]
        for u, v in zip(a[1:], b[1:]):
            assert u == pytest.approx(v, rel=2e-8, abs=1e-6)
    m = re.search(r"fix nve/mdp: (\d+) bricks, (\d+) reneighborings on the device, (\d+) returns of the atoms", out)
    assert m and int(m.group(1)) == np and int(m.group(2)) >= 3 and int(m.group(3)) == 4
    counts = [int(x) for x in re.findall(r"rank \d+: Nlocal (\d+)", out)]
    assert len(counts) == np and sum(counts) == 32000 and len(set(counts)) > 1   # (what came back is what the bricks own now)


def test_fix_nve_mdp_on_four_ranks_continues_over_two_runs():
    """run 50 + run 50: at the end of a run the atoms are the host's again (on the ranks they migrated to), the second
    run's setup is the host's own (exchange, borders, lists, host-mode forces) and hands them to the bricks anew"""
    rc1, out1, err1 = _run(_aeam_mdp(), np=4, env=_double_env())
    assert rc1 == 0, err1
    rc, out, err = _run(_aeam_mdp("run 50\nrun 50"), np=4, env=_double_env())
    assert rc == 0, err
    whole, parts = _thermo_rows(out1), _thermo_rows(out)
    assert [int(r[0]) for r in parts] == [0, 25, 50, 50, 75, 100]
    for a, b in zip(parts[:3] + parts[4:], whole):
        for u, v in zip(a, b):
            assert u == pytest.approx(v, rel=2e-8, abs=1e-6)


def test_fix_nve_mdp_rebomos_hot_on_four_ranks_equals_the_hosts_fix_nve():
    """2 304 atoms of MoS2 from 1 500 K with 0.4 A of skin, 300 steps: many reneighborings on the bricks; thermo rows of
    the host's own `fix nve` on one rank"""
    from test_gpu_fix_nve_mdp import REBO_HOT, _script
    base = _script("in.rebomos-bulk.mi355x", **REBO_HOT)
    rc0, out0, err0 = _run(base)
    assert rc0 == 0, err0
    rc, out, err = _run(base.replace("fix integrate all nve", "fix integrate all nve/mdp"), np=4, env=_double_env())
    assert rc == 0, err
    r0, r1 = _thermo_rows(out0), _thermo_rows(out)
    assert len(r0) == len(r1) == 7
    for a, b in zip(r1, r0):
        for u, v in zip(a, b):
            assert u == pytest.approx(v, rel=5e-7, abs=1e-5)
    m = re.search(r"fix nve/mdp: 4 bricks, (\d+) reneighborings on the device", out)
    assert m and int(m.group(1)) > 5


def test_rebomos_hot_host_mode_on_four_ranks_equals_one_rank():
    """the same hot MoS2 cell in HOST MODE (the host's own fix nve, exchange and borders at its many reneighborings, ghosts
    of other ranks uploaded with the owned atoms every step): rows of the one-rank run"""
    from test_gpu_fix_nve_mdp import REBO_HOT, _script
    base = _script("in.rebomos-bulk.mi355x", **REBO_HOT)
    rc0, out0, err0 = _run(base)
    assert rc0 == 0, err0
    rc, out, err = _run(base, np=4)
    assert rc == 0, err
    r0, r1 = _thermo_rows(out0), _thermo_rows(out)
    assert len(r0) == len(r1) == 7
    for a, b in zip(r1, r0):
        for u, v in zip(a, b):
            assert u == pytest.approx(v, rel=5e-7, abs=1e-5)
    assert re.search(r"Neighbor list builds = (\d+)", out).group(1) == re.search(r"Neighbor list builds = (\d+)", out0).group(1)
    counts = [int(x) for x in re.findall(r"rank \d+: Nlocal (\d+)", out)]
    assert sum(counts) == 2304
