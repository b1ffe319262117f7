"""CPU: the oracle still reproduces the frozen Appendix-C fixtures (tests/golden/fixtures/, written by
tests/make_fixtures.py).  The fixtures were made BY this oracle: the test pins nothing to the reference (that is
tests/test_oracle_rebomos.py, log.rebomos-bulk.1:54-56); it makes a silent edit of the oracle -- alone, or together with
the kernels it checks -- fail.  Equality is exact on the machine that wrote the files; a host whose libm dispatches other
exp / sin / cos variants (FMA or not) may differ in the last bits, so the bound is 1e-12 of each quantity's scale, five
orders below every tolerance of the GPU parity tests."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, POT_AEAM, POT_REBOMOS
import fixture_cases as FC

TIGHT = 1e-12


@pytest.fixture(scope="module")
def pots(oracle):
    return oracle.rebomos_params(POT_REBOMOS), oracle.aeam_pot(POT_AEAM)


def test_the_fixture_set_is_complete():
    have = sorted(f[:-4] for f in os.listdir(FC.FIXDIR) if f.endswith(".npz"))
    assert have == FC.names()


@pytest.mark.parametrize("name", FC.names())
def test_oracle_reproduces_fixture(oracle, pots, name):
    style, s, want, sample, sums = FC.load(name)
    eng = FC.engine(style, s, oracle, P=pots[0], T=pots[1])
    got = FC.oracle_outputs(style, eng, s.x)
    for k, w in want.items():
        g = got[k]
        if sample is not None and k in FC.PER_ATOM:
            assert np.abs(np.asarray(g, dtype=np.float64).sum(axis=0) - sums[k]).max() <= TIGHT * max(1.0, np.abs(sums[k]).max()) * s.n
            g = g[sample]
        if w.dtype.kind in "iu":
            assert np.array_equal(g, w), k
        else:
            scale = max(1.0, float(np.abs(w).max())) if w.size else 1.0
            assert np.abs(np.asarray(g) - w).max() <= TIGHT * scale if w.size else True, k


def test_bulk_fixtures_are_the_reference_log():
    """the three R-bulk-0 states carry the thermo rows of log.rebomos-bulk.1:54-56 (PE to the printed digits)"""
    log = json.load(open(os.path.join(GOLDEN, "rebomos_bulk_log.json")))
    for row, step in zip(log["thermo"], FC.BULK_STEPS):
        _, s, want, _, _ = FC.load(f"R-bulk-0-step{step}")
        assert s.n == 288
        assert float(want["eng"]) == pytest.approx(row["pe"], abs=5.1e-5)
    _, s8, w8, _, _ = FC.load("R-repl-2")
    assert float(w8["eng"]) == pytest.approx(8 * log["thermo"][0]["pe"], abs=4.1e-4)   # replicate KAT


def test_fixture_branch_coverage():
    """what Appendix C says each case is for, checked on the stored numbers: switching interior (fractional nM / nS),
    the AEAM eatom quirk (sum eatom != PE with angular atoms, pair_aeam.cpp:294-300), isolated-Si clamp (rho = 0)"""
    for name in ("R-strain-112", "R-comp-093", "R-jit-100"):
        _, s, w, _, _ = FC.load(name)
        n = w["nM"] + w["nS"]
        assert np.any(np.abs(n - np.round(n)) > 1e-3), name         # some bond sits inside a switching interval
        assert float(w["eatom"].sum()) == pytest.approx(float(w["eng"]), rel=1e-11)
    _, s, w, _, _ = FC.load("R-bulk-0-step0")
    n = w["nM"] + w["nS"]
    assert np.all(n == np.round(n))                                 # the log's cell: every bond at t <= 0
    assert sorted(set(w["rebo_numneigh"][s.type == 1])) == [12] and sorted(set(w["rebo_numneigh"][s.type == 2])) == [3]
    _, s, w, _, _ = FC.load("A-6-8pct")
    assert abs(float(w["eatom"].sum()) - float(w["eng"])) > 1.0
    _, s, w, _, _ = FC.load("A-edge")
    assert w["rho"][0] == 0.0 and s.type[0] == 2 and not w["f"][0].any()
