"""Pin the REBO-MoS CPU oracle to the reference's only known-answer data
(USER-REBOMOS/log.rebomos-bulk.1:54-56,72-83) and check its internal consistency on the
branches that log never reaches (switching-function interior, LJ cubic)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, POT_REBOMOS
from lammps_plugins_amd.host import system as S
import mdref


@pytest.fixture(scope="module")
def log():
    with open(os.path.join(GOLDEN, "rebomos_bulk_log.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="module")
def P(oracle):
    return oracle.rebomos_params(POT_REBOMOS)


def test_params_mixing(P):
    # pair_rebomos.cpp:1048-1066, 262-265
    assert P.sigma[0][1] == pytest.approx((4.2 + 3.13) / 2)
    assert P.epsilon[0][1] == pytest.approx(np.sqrt(0.00058595 * 0.01386))
    assert P.rcLJmax[0][0] == pytest.approx(10.5) and P.rcLJmax[1][1] == pytest.approx(7.825)
    assert P.rcLJmin[0][1] == 2.75 and P.cut3rebo == pytest.approx(11.4)
    assert P.lj3[1][1] == pytest.approx(4 * 0.01386 * 3.13 ** 12)
    assert P.b[6][0] == -0.282577591351457 and P.bg[0][1] == -0.2850852 and P.a[3][1] == 2.386431372486710


def test_cell_matches_log_geometry(log):
    s = S.rebomos_bulk_cell()
    assert s.n == log["natoms"]
    assert np.allclose(s.box.prd, log["box_hi"], atol=5e-7)
    assert s.box.tilt[0] == pytest.approx(log["tilt_xy"], abs=5e-8)
    assert s.box.volume == pytest.approx(log["volume"], abs=5e-5)
    assert int((s.type == 1).sum()) == 96 and int((s.type == 2).sum()) == 192


def test_thermo_table_matches_reference_log(oracle, P, log):
    """20 NVE steps from rest: every printed digit of the reference log."""
    s = S.rebomos_bulk_cell()
    eng = mdref.RebomosCPU(oracle, P, s)
    assert eng.nghost == log["nghost"]
    assert int(eng.nn[:eng.nlocal].sum()) == log["full_neighbors"]
    rows, _, _ = mdref.nve(eng, s, 20)
    for got, ref in zip(rows, log["thermo"]):
        assert got["step"] == ref["step"]
        assert got["pe"] == pytest.approx(ref["pe"], abs=5.1e-5)        # printed to 1e-4
        assert got["ke"] == pytest.approx(ref["ke"], abs=5.1e-8)
        assert got["temp"] == pytest.approx(ref["temp"], abs=5.1e-6 if ref["temp"] < 100 else 5.1e-6 * 10)
        assert got["press"] == pytest.approx(ref["press"], abs=5.1e-3)


def test_replicated_cell_energy_is_extensive(oracle, P, log):
    s = S.replicate(S.rebomos_bulk_cell(), (2, 1, 1))
    o = mdref.RebomosCPU(oracle, P, s).compute(s.x)
    assert o["eng"] == pytest.approx(2 * log["thermo"][0]["pe"], abs=2e-4)
    assert np.abs(o["f_owned"].sum(axis=0)).max() < 1e-10


def _fd_check(oracle, P, s, atoms, h=1e-5):
    eng = mdref.RebomosCPU(oracle, P, s)
    o = eng.compute(s.x)
    f = o["f_owned"]
    worst = 0.0
    for a in atoms:
        for d in range(3):
            xp = s.x.copy()
            xp[a, d] += h
            xm = s.x.copy()
            xm[a, d] -= h
            fd = -(eng.compute(xp)["eng"] - eng.compute(xm)["eng"]) / (2 * h)
            worst = max(worst, abs(fd - f[a, d]))
    return worst, o, eng


@pytest.mark.parametrize("fac,amp", [(1.12, 0.15), (0.93, 0.10), (1.0, 0.15)])
def test_forces_are_energy_gradient_on_all_branches(oracle, P, fac, amp):
    """strained + jittered cells enter the switching interior and the LJ cubic branch
    (SURVEY.md Appendix C R-strain-112 / R-comp-093 / R-jit-100)."""
    s = S.jitter(S.scale(S.rebomos_bulk_cell(), fac), amp, seed=1234)
    worst, o, eng = _fd_check(oracle, P, s, atoms=[0, 5, 17, 100, 287])
    assert worst < 2e-6
    assert np.abs(o["f_owned"].sum(axis=0)).max() < 1e-9
    # explicit tally virial == fdotr virial; sum of eatom == PE
    assert np.allclose(o["virial_tally"], o["virial_fdotr"], rtol=1e-9, atol=1e-8)
    assert o["eatom"].sum() == pytest.approx(o["eng"], rel=1e-12)
    # per-atom virial sums to the global one
    assert np.allclose(o["vatom"].sum(axis=0), o["virial_tally"], rtol=1e-9, atol=1e-8)


def test_branch_coverage_of_strained_cell(oracle, P):
    """make sure the strained fixture really reaches the branches the log does not"""
    s = S.jitter(S.scale(S.rebomos_bulk_cell(), 1.12), 0.15, seed=1234)
    eng = mdref.RebomosCPU(oracle, P, s)
    xa = eng.all_positions(s.x)
    i = np.repeat(np.arange(eng.nlocal), eng.nn[:eng.nlocal])
    j = eng.nb[:eng.off[eng.nlocal]]
    r = np.linalg.norm(xa[i] - xa[j], axis=1)
    ti, tj = eng.elem[i], eng.elem[j]
    rcmin = np.array([[P.rcmin[a][b] for b in range(2)] for a in range(2)])[ti, tj]
    rcmax = np.array([[P.rcmax[a][b] for b in range(2)] for a in range(2)])[ti, tj]
    sig = np.array([[P.sigma[a][b] for b in range(2)] for a in range(2)])[ti, tj]
    assert ((r > rcmin) & (r < rcmax)).sum() > 100          # switching interior
    assert ((r >= rcmin) & (r < 0.95 * sig)).sum() > 100    # LJ cubic branch
