"""GPU, BASELINE.json full size: REBO-MoS bulk replicated 24x24x24 = 3,981,312 atoms (config #4) and
AEAM 63^3 x 4 = 1,000,188 atoms (config #3), checked through size-independent properties: the known
answers of the 288-atom cell scale extensively (PE = 13824 x (-2061.6112) eV, PE/atom =
-7.158372 eV, P = 28799.53 bar), momentum conservation, NVE energy conservation, and the PE/KE of the
replicated system after 10 steps equal the log's step-10 row times 13824 (every replica moves alike)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, POT_AEAM, POT_REBOMOS
from lammps_plugins_amd.host import capi, resident, system as S

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_rebomos_4m_atoms_known_answers():
    log = json.load(open(os.path.join(GOLDEN, "rebomos_bulk_log.json")))
    nrep = 24 ** 3
    s = S.replicate(S.rebomos_bulk_cell(), (24, 24, 24))
    assert s.n == 3981312
    ctx = capi.Context(0)
    p = capi.read_rebomos_file(POT_REBOMOS)
    ctx.rebomos_set_params(p)
    d = resident.make_domain(ctx, capi.STYLE_REBOMOS, s, 3.0 * p.rcmax[0][0] + 2.0, 2.0, [0, 0, 1])
    d.build_neighbors()
    d.compute(eflag=1, vflag=1)
    t0 = d.thermo()
    ref0, ref10 = log["thermo"][0], log["thermo"][1]
    assert t0["pe"] / nrep == pytest.approx(ref0["pe"], abs=5.1e-5)
    assert t0["pe"] / s.n == pytest.approx(-7.158372, abs=5e-7)
    assert t0["press"] == pytest.approx(ref0["press"], abs=5.1e-3)
    f = ctx.md_download(s.n, want=("f",))["f"]
    assert np.abs(f.sum(axis=0)).max() < 1e-6              # 4M-term sums
    assert np.abs(f).max() < 10.0
    for step in range(1, 11):
        ev = 1 if step == 10 else 0
        d.step(ev, ev)
    t10 = d.thermo()
    assert t10["pe"] / nrep == pytest.approx(ref10["pe"], abs=5.1e-5)
    assert t10["ke"] / nrep == pytest.approx(ref10["ke"], abs=5.1e-8)
    # the 13824 replicas share 3 degrees of freedom fewer than 13824 separate cells: compare energies, and
    # the pressure through its definition
    assert (t10["pe"] + t10["ke"]) == pytest.approx(t0["pe"] + t0["ke"], abs=2e-5 * s.n)
    ctx.close()


@pytest.mark.timeout(900)
def test_aeam_1m_atoms_properties():
    af = capi.AeamFile(POT_AEAM)
    tabs = af.build()
    ctx = capi.Context(0)
    ctx.aeam_set_tables(tabs)
    s = S.fcc_cell(4.045, (63, 63, 63), frac_type2=0.0075, seed=7683797)
    assert s.n == 1000188
    s.mass[1:3] = af.mass
    v0 = S.gaussian_velocities(s, 863.0, seed=1082337)
    cutghost = float(af.cut_table(tabs).max()) + 1.0
    d = resident.make_domain(ctx, capi.STYLE_AEAM, s, cutghost, 1.0, None, v0=v0)
    d.build_neighbors()
    assert ctx.md_neighbor_stats()[0] / s.n == pytest.approx(85.35, abs=0.3)   # 85.35 entries/atom
    d.compute(eflag=1, vflag=1)
    t0 = d.thermo()
    assert -3.43 < t0["pe"] / s.n < -3.40            # perfect lattice with 0.75 % Si (~ -3.4122)
    assert t0["temp"] == pytest.approx(863.0, rel=1e-9)
    f = ctx.md_download(s.n, want=("f",))["f"]
    assert np.abs(f.sum(axis=0)).max() < 1e-7
    e0 = t0["pe"] + t0["ke"]
    for step in range(1, 41):
        d.ctx.md_initial_integrate()
        if step % 10 == 0 and d.needs_rebuild():
            d = resident.reneighbor(d, s, cutghost, None)
        d.compute(0, 0)
        d.ctx.md_final_integrate()
    d.compute(eflag=1, vflag=0)
    t1 = d.thermo()
    # velocity-Verlet at 863 K from a perfect lattice, dt = 1 fs: O((w dt)^2 KE) ~ 1e-4 eV/atom fluctuation
    assert abs(t1["pe"] + t1["ke"] - e0) / s.n < 1.5e-4
    ctx.close()
