"""The product's AEAM potential-file front end (csrc/potfile.cpp: mdp_aeam_file_read / _build) checked
WITHOUT the oracle: the file is parsed again here in Python and the spline rows are restated with whole-array
numpy expressions (a third, structurally different formulation of USER-AEAM/pair_aeam.cpp:915-942), then
selected rows -- both ends, the rows next to the ends, an interior row -- are also written out by hand from the
raw tabulated values.  Bit equality is required: every coefficient is one correctly rounded expression of the
file's numbers, so any change of operand order in the product shows up here.

Runs on the CPU (the library only has to load; no device call is made)."""
import numpy as np
import pytest

from conftest import POT_AEAM
from lammps_plugins_amd.host import capi


def _parse(path):
    """AlSi.aeam layout (pair_aeam.cpp:645-746): 12 header lines, element line, per element `nrho drho mass`,
    per ordered element pair `nr dr cut`, then F(rho) per element, rho(r) per ordered pair, phi(r) per pair i>=j"""
    with open(path) as fh:
        lines = fh.read().split("\n")
    tok = lines[11].split()
    ne = int(tok[0])
    pos = 12
    nrho, drho = [], []
    for _ in range(ne):
        t = lines[pos].split()
        pos += 1
        nrho.append(int(t[0]))
        drho.append(float(t[1]))
    nr, dr = [], []
    for _ in range(ne * ne):
        t = lines[pos].split()
        pos += 1
        nr.append(int(t[0]))
        dr.append(float(t[1]))
    vals = np.array(" ".join(l.split("#")[0] for l in lines[pos:]).split(), dtype=np.float64)
    out, at = {"F": [], "rho": [], "phi": []}, 0
    for i in range(ne):
        out["F"].append(vals[at:at + nrho[i]])
        at += nrho[i]
    for k in range(ne * ne):
        out["rho"].append(vals[at:at + nr[k]])
        at += nr[k]
    for i in range(ne):
        for j in range(i + 1):
            n = nr[i * ne + j]
            out["phi"].append(vals[at:at + n])
            at += n
    assert at == len(vals)
    return ne, nrho, drho, nr, dr, out


def _rows_numpy(y, h):
    """rows 1..n as an (n+1, 7) array (row 0 unused, zero): whole-array restatement"""
    n = len(y)
    f = np.concatenate([[0.0], y])            # 1-based
    s = np.zeros(n + 1)
    m = np.arange(3, n - 1)
    s[3:n - 1] = ((f[m - 2] - f[m + 2]) + 8.0 * (f[m + 1] - f[m - 1])) / 12.0
    s[1] = f[2] - f[1]
    s[2] = 0.5 * (f[3] - f[1])
    s[n - 1] = 0.5 * (f[n] - f[n - 2])
    s[n] = f[n] - f[n - 1]
    rise = np.zeros(n + 1)
    rise[1:n] = f[2:n + 1] - f[1:n]
    snext = np.zeros(n + 1)
    snext[1:n] = s[2:n + 1]
    quad = 3.0 * rise - 2.0 * s - snext
    cubic = s + snext - 2.0 * rise
    quad[n] = cubic[n] = 0.0
    quad[0] = cubic[0] = 0.0
    rows = np.zeros((n + 1, 7))
    rows[:, 6], rows[:, 5], rows[:, 4], rows[:, 3] = f, s, quad, cubic
    rows[:, 2], rows[:, 1], rows[:, 0] = s / h, 2.0 * quad / h, 3.0 * cubic / h
    rows[0] = 0.0
    return rows


@pytest.fixture(scope="module")
def built():
    af = capi.AeamFile(POT_AEAM)
    tabs = af.build()
    return af, tabs


def _table(ptr, ntab, nmax):
    return np.ctypeslib.as_array(ptr, shape=(ntab, nmax + 1, 7))


def test_every_row_equals_the_numpy_restatement(built):
    af, tabs = built
    ne, nrho, drho, nr, dr, raw = _parse(POT_AEAM)
    assert (tabs.nelements, tabs.nfrho, tabs.nrhor, tabs.nz2r) == (ne, ne + 1, ne * ne, ne * (ne + 1) // 2)
    fr = _table(tabs.frho_spline, tabs.nfrho, tabs.nrhomax)
    rh = _table(tabs.rhor_spline, tabs.nrhor, tabs.nrmax)
    z2 = _table(tabs.z2r_spline, tabs.nz2r, tabs.nrmax)
    for i in range(ne):
        assert np.array_equal(fr[i][:nrho[i] + 1], _rows_numpy(raw["F"][i], drho[i]))
    assert np.all(fr[ne] == 0.0)          # the extra all-zero table for NULL types (pair_aeam.cpp:767-779)
    for k in range(ne * ne):
        assert np.array_equal(rh[k][:nr[k] + 1], _rows_numpy(raw["rho"][k], dr[k]))
    t = 0
    for i in range(ne):
        for j in range(i + 1):
            k = i * ne + j
            assert np.array_equal(z2[t][:nr[k] + 1], _rows_numpy(raw["phi"][t], dr[k]))
            t += 1


def test_hand_written_rows(built):
    """rows 1, 2, an interior row, n-1 and n of the Al-Al density table, written out from the file's numbers"""
    _, tabs = built
    ne, nrho, drho, nr, dr, raw = _parse(POT_AEAM)
    y, h, n = raw["rho"][0], dr[0], nr[0]
    f = lambda m: y[m - 1]                                         # 1-based tabulated value
    rh = _table(tabs.rhor_spline, tabs.nrhor, tabs.nrmax)[0]

    def check(m, slope, slope_next):
        rise = f(m + 1) - f(m) if m < n else 0.0
        quad = 3.0 * rise - 2.0 * slope - slope_next if m < n else 0.0
        cubic = slope + slope_next - 2.0 * rise if m < n else 0.0
        want = [3.0 * cubic / h, 2.0 * quad / h, slope / h, cubic, quad, slope, f(m)]
        assert list(rh[m]) == want, m

    five = lambda m: ((f(m - 2) - f(m + 2)) + 8.0 * (f(m + 1) - f(m - 1))) / 12.0
    check(1, f(2) - f(1), 0.5 * (f(3) - f(1)))
    check(2, 0.5 * (f(3) - f(1)), five(3))
    check(4321, five(4321), five(4322))
    check(n - 2, five(n - 2), 0.5 * (f(n) - f(n - 2)))
    check(n - 1, 0.5 * (f(n) - f(n - 2)), f(n) - f(n - 1))
    check(n, f(n) - f(n - 1), 0.0)


def test_type_maps_and_null_types():
    """type -> table maps (pair_aeam.cpp:785-871), incl. a NULL-mapped type and swapped element order"""
    af = capi.AeamFile(POT_AEAM)
    t = af.build()
    nt = t.ntypes
    r = np.ctypeslib.as_array(t.type2rhor, shape=(nt + 1, nt + 1))
    z = np.ctypeslib.as_array(t.type2z2r, shape=(nt + 1, nt + 1))
    fmap = np.ctypeslib.as_array(t.type2frho, shape=(nt + 1,))
    assert r[1:, 1:].tolist() == [[0, 1], [2, 3]]
    assert z[1:, 1:].tolist() == [[0, 1], [1, 2]]
    assert fmap[1:].tolist() == [0, 1]
    af2 = capi.AeamFile(POT_AEAM)
    t2 = af2.build(3, map_=[0, 1, -1, 0])                          # types: Si, NULL, Al
    r2 = np.ctypeslib.as_array(t2.type2rhor, shape=(4, 4))
    z2 = np.ctypeslib.as_array(t2.type2z2r, shape=(4, 4))
    f2 = np.ctypeslib.as_array(t2.type2frho, shape=(4,))
    assert r2[1:, 1:].tolist() == [[0, 1, 2], [3, 4, 5], [6, 7, 8]]   # numbered by TYPE pair, stride ntypes
    assert z2[1:, 1:].tolist() == [[2, 0, 1], [0, 0, 0], [1, 0, 0]]   # Si-Si=2, Si-Al=Al-Si=1, Al-Al=0, NULL -> 0
    assert f2[1:].tolist() == [1, t2.nfrho - 1, 0]
