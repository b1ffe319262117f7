"""GPU parity against the FROZEN fixtures of SURVEY Appendix C (tests/golden/fixtures/*.npz): the HIP path through the
C-ABI on the stored inputs, compared with the stored outputs -- no oracle call in this file.  The other -m gpu files
compare with a live oracle built on the GPU box; here an edit that changed oracle and kernel alike would still fail.
Tolerances as everywhere: forces 1e-9 eV/A, per-atom energy 1e-9 eV (north star 1e-6), PE 1e-10 rel, virial 1e-9 rel."""
import numpy as np
import pytest

from conftest import POT_AEAM, POT_REBOMOS
from lammps_plugins_amd.host import capi
import fixture_cases as FC
import oracle_bindings as ob

pytestmark = pytest.mark.gpu

REBO = [n for n in FC.names() if n.startswith("R-")]
AEAM = [n for n in FC.names() if n.startswith("A-")]


def _fold(a, owner, nlocal):
    out = a[:nlocal].copy()
    np.add.at(out, owner, a[nlocal:])
    return out


@pytest.fixture(scope="module")
def rctx():
    c = capi.Context(0)
    c.params = capi.read_rebomos_file(POT_REBOMOS)
    c.rebomos_set_params(c.params)
    yield c
    c.close()


@pytest.mark.parametrize("lists", ["device", "host_csr"])
@pytest.mark.parametrize("name", REBO)
def test_rebomos_fixture(rctx, name, lists):
    style, s, want, sample, _ = FC.load(name)
    rcmax = [[rctx.params.rcmax[a][b] for b in range(2)] for a in range(2)]
    eng = FC.lists_only(style, s, rcmax)                  # ghosts + lists only; nothing here calls the oracle
    xa = eng.all_positions(s.x)
    rctx.set_atoms_host(eng.nlocal, xa, eng.type_all, eng.tag_all, 2, map_=[0, 0, 1])
    if lists == "device":
        rctx.set_skin(2.0)
    else:
        rctx.set_neighbors_csr_host(eng.nn, eng.off, eng.nb, 2.0)
    g = rctx.rebomos_compute_host(eng.nlocal, eflag=3, vflag=5)
    assert np.abs(g["f"] - want["f"]).max() < 1e-9
    assert g["eng"] == pytest.approx(float(want["eng"]), rel=1e-10)
    assert np.abs(g["eatom"] - want["eatom"]).max() < 1e-9
    assert np.allclose(g["virial"], want["virial_fdotr"], rtol=1e-9, atol=1e-7)
    assert np.abs(g["vatom"] - want["vatom"]).max() < 1e-9 * max(1.0, np.abs(want["vatom"]).max())
    # a force-only call (the kernels an MD step runs) on the same inputs
    g0 = rctx.rebomos_compute_host(eng.nlocal, eflag=0, vflag=0)
    assert np.abs(g0["f"] - want["f"]).max() < 1e-9


@pytest.mark.parametrize("lists", ["device", "host_csr"])
@pytest.mark.parametrize("name", AEAM)
def test_aeam_fixture(name, lists):
    style, s, want, sample, sums = FC.load(name)
    if lists == "host_csr" and s.n > FC.LARGE:
        pytest.skip("the 32 000-atom cases go through the device-built lists (the drop-in default)")
    af = capi.AeamFile(POT_AEAM)
    tabs = af.build()
    ctx = capi.Context(0)
    ctx.aeam_set_tables(tabs)
    if lists == "device":
        import lammps_plugins_amd.host.system as S
        cut = float(af.cut_table(tabs).max()) + 1.0
        xa, type_all, tag_all, owner, _, nloc, _ = S.with_ghosts(s, cut)
        ctx.aeam_device_lists(True)
        ctx.set_atoms_host(nloc, xa, type_all, tag_all, 2, map_=None)
        ctx.set_skin(1.0)
    else:
        cut = af.cut_table(tabs)[1:, 1:]
        eng = FC.lists_only(style, s, [[float(cut[a][b]) for b in range(2)] for a in range(2)])
        xa, owner, nloc = eng.all_positions(s.x), eng.owner, eng.nlocal
        ctx.set_atoms_host(nloc, xa, eng.type_all, eng.tag_all, 2, map_=None)
        ctx.set_neighbors_csr_host(eng.nn, eng.off, eng.nb, 1.0)
    nall = len(xa)
    d = ctx.aeam_density_host(nloc, eflag=3)
    fp_all = np.concatenate([d["fp"], d["fp"][owner]])              # forward_comm on one periodic rank (pair_aeam.cpp:307)
    r = ctx.aeam_force_host(nall, nloc, fp_all, eflag=3, vflag=5)
    got = dict(f=ob.fold_ghost_forces(r["f"], owner, nloc), eatom=d["eatom"] + r["eatom"], rho=d["rho"], fp=d["fp"],
               vatom=_fold(r["vatom"], owner, nloc))
    pe = d["eng"] + r["eng"]
    fscale = max(1.0, float(np.abs(want["f"]).max()))
    pick = (lambda a: a[sample]) if sample is not None else (lambda a: a)
    assert np.abs(pick(got["rho"]) - want["rho"]).max() < 1e-11 * max(1.0, np.abs(want["rho"]).max())
    # what the library hands the host's forward_comm is q = Fptmp * F' (the two factors only ever appear as a product,
    # pair_aeam.cpp:329-332, 373, 450-452): F' itself for a metal, F' / (2 sqrt(rho)) for an angular atom (0 at rho = 0)
    ty, rho = pick(s.type), want["rho"]
    fptmp = np.where(rho > 1e-13, np.where(ty <= af.nnonangular, 1.0, 0.5 / np.sqrt(np.maximum(rho, 1e-300))), 0.0)
    assert np.abs(pick(got["fp"]) - want["fp"] * fptmp).max() < 1e-10 * max(1.0, np.abs(want["fp"] * fptmp).max())
    assert np.abs(pick(got["f"]) - want["f"]).max() < 1e-9 * fscale
    assert np.abs(pick(got["eatom"]) - want["eatom"]).max() < 1e-9
    assert np.abs(pick(got["vatom"]) - want["vatom"]).max() < 1e-9 * max(1.0, np.abs(want["vatom"]).max())
    assert pe == pytest.approx(float(want["eng"]), rel=1e-10)          # (32 000 terms summed in another order)
    assert np.allclose(r["virial"], want["virial_fdotr"], rtol=1e-9, atol=1e-7 * fscale)
    if sample is not None:                                          # the unsampled atoms through the stored sums
        assert np.abs(got["f"].sum(axis=0) - sums["f"]).max() < 1e-7
        assert got["eatom"].sum() == pytest.approx(float(sums["eatom"]), rel=1e-11)
        assert got["rho"].sum() == pytest.approx(float(sums["rho"]), rel=1e-11)
    # force-only call
    d0 = ctx.aeam_density_host(nloc, eflag=0)
    r0 = ctx.aeam_force_host(nall, nloc, np.concatenate([d0["fp"], d0["fp"][owner]]), eflag=0, vflag=0)
    assert np.abs(pick(ob.fold_ghost_forces(r0["f"], owner, nloc)) - want["f"]).max() < 1e-9 * fscale
    ctx.close()
