"""Error paths of the C-ABI (include/mdpair_hip.h): call-order errors (MDP_ESTATE), neighbor overflow
(MDP_EOVERFLOW, the reference's "Neighbor list overflow, boost neigh_modify one", pair_rebomos.cpp:350), damaged
potential files (pair_aeam.cpp:665,684,709; pair_rebomos.cpp:955-957), empty sub-domains handed over as real NULL
pointers, and the guard against host lists that are not the plain geometric list.

CPU part: the potential-file front ends are host code.  GPU part (marked): everything that needs a context."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import POT_AEAM, POT_REBOMOS
from lammps_plugins_amd.host import capi, system as S
import hostplan

MDP_EINVAL, MDP_EOVERFLOW, MDP_ESTATE = -1, -4, -6


# ------------------------------------------------------------------------------------------ CPU: bad files
def _aeam_err(path):
    with pytest.raises(capi.MdpError) as e:
        capi.AeamFile(str(path))
    return str(e.value)


def test_truncated_aeam_file_is_reported(tmp_path):
    lines = open(POT_AEAM).read().split("\n")
    p = tmp_path / "short.aeam"
    p.write_text("\n".join(lines[:2000]) + "\n")          # ends inside F(rho) of the first element
    msg = _aeam_err(p)
    assert "AEAM potential file parser error" in msg and "end of file" in msg
    p.write_text("\n".join(lines[:5]) + "\n")              # ends inside the 12-line header
    assert "AEAM potential file parser error" in _aeam_err(p)
    p.write_text("\n".join(lines[:14]) + "\n")             # element line + one of two nrho/drho/mass lines
    assert "AEAM potential file parser error" in _aeam_err(p)


def test_non_numeric_aeam_header_is_reported(tmp_path):
    lines = open(POT_AEAM).read().split("\n")
    bad = list(lines)
    bad[11] = "two 1 1 Al Si"                              # element-count line (pair_aeam.cpp:655-666)
    p = tmp_path / "bad.aeam"
    p.write_text("\n".join(bad))
    msg = _aeam_err(p)
    assert "AEAM potential file parser error" in msg and "Not a valid integer" in msg
    bad = list(lines)
    bad[12] = "  10000 abc 27 Al"                          # nrho drho mass (pair_aeam.cpp:676-686)
    p.write_text("\n".join(bad))
    assert "AEAM potential file parser error" in _aeam_err(p)
    assert "Cannot open AEAM potential file" in _aeam_err(tmp_path / "missing.aeam")


def test_bad_rebomos_files_are_reported(tmp_path):
    lines = open(POT_REBOMOS).read().split("\n")
    data = [k for k, l in enumerate(lines) if l.strip() and not l.strip().startswith("#")]
    bad = list(lines)
    bad[data[7]] = "3.x5   # not a number"
    p = tmp_path / "bad.set5b"
    p.write_text("\n".join(bad))
    with pytest.raises(capi.MdpError) as e:
        capi.read_rebomos_file(str(p))
    msg = str(e.value)                                     # message shape of pair_rebomos.cpp:955
    assert "reading rebomos potential file" in msg and "REASON: Not a valid floating-point number: '3.x5'" in msg
    p.write_text("\n".join(lines[:data[40]]) + "\n")       # 40 of 61 scalars
    with pytest.raises(capi.MdpError) as e:
        capi.read_rebomos_file(str(p))
    assert "REASON: unexpected end of file" in str(e.value)
    with pytest.raises(capi.MdpError) as e:
        capi.read_rebomos_file(str(tmp_path / "missing.set5b"))
    assert "cannot open rebomos potential file" in str(e.value)


# ------------------------------------------------------------------------------------------ GPU: context errors
gpu = pytest.mark.gpu
BOX = S.Box(np.zeros(3), np.array([60.0, 60.0, 60.0]), np.zeros(3))


def _code(fn, *a, **k):
    with pytest.raises(capi.MdpError) as e:
        fn(*a, **k)
    return e.value.code, str(e.value)


def _sphere(n, radius, seed=5):
    """n points spread over a ball (golden-spiral shells), pairwise distinct"""
    k = np.arange(n) + 0.5
    phi = np.arccos(1 - 2 * k / n)
    th = np.pi * (1 + 5 ** 0.5) * k
    r = radius * ((k / n) ** (1.0 / 3.0))
    return np.stack([r * np.cos(th) * np.sin(phi), r * np.sin(th) * np.sin(phi), r * np.cos(phi)], axis=1)


@gpu
def test_calls_before_their_prerequisites_return_ESTATE():
    ctx = capi.Context(0)
    x = np.zeros((2, 3))
    x[1, 0] = 2.4
    assert _code(ctx.rebomos_compute_host, 2)[0] == MDP_ESTATE          # no parameters yet
    assert _code(ctx.set_positions_host, x)[0] == MDP_ESTATE            # no atoms yet
    assert _code(ctx.aeam_density_host, 2)[0] == MDP_ESTATE             # no tables yet
    assert _code(ctx.md_compute)[0] == MDP_ESTATE                       # no mdp_md_setup yet
    assert _code(ctx.md_build_neighbors)[0] == MDP_ESTATE
    assert _code(ctx.md_thermo)[0] == MDP_ESTATE
    ctx.rebomos_set_params(capi.read_rebomos_file(POT_REBOMOS))
    code, msg = _code(ctx.rebomos_compute_host, 2)
    assert code == MDP_ESTATE and "atoms not set" in msg
    assert _code(ctx.rebomos_list_info)[0] == MDP_ESTATE                # lists not built yet
    af = capi.AeamFile(POT_AEAM)
    ctx.aeam_set_tables(af.build())
    ctx.set_atoms_host(2, x + 30.0, np.array([1, 1], np.int32), np.array([1, 2], np.int32), 2)
    code, msg = _code(ctx.aeam_density_host, 2)
    assert code == MDP_ESTATE and "neighbor list not set" in msg
    ctx.close()


@gpu
def test_more_than_64_rebo_candidates_is_a_neighbor_list_overflow():
    """70 S atoms inside a 1.9 A ball: every atom has 69 candidates within rcmax+skin -- beyond the 64 slots of
    the active mask, as `oneatom` bounds the reference's REBO pages (pair_rebomos.cpp:345-350)"""
    ctx = capi.Context(0)
    ctx.rebomos_set_params(capi.read_rebomos_file(POT_REBOMOS))
    x = _sphere(70, 1.9) + 30.0
    ctx.set_atoms_host(70, x, np.full(70, 2, np.int32), np.arange(1, 71, dtype=np.int32), 2, map_=[0, 0, 1])
    ctx.set_skin(2.0)
    code, msg = _code(ctx.rebomos_compute_host, 70)
    assert code == MDP_EOVERFLOW and "Neighbor list overflow" in msg
    # the context stays usable: a sane system afterwards computes
    x2 = np.array([[30.0, 30, 30], [32.41, 30, 30]])
    ctx.set_atoms_host(2, x2, np.array([1, 2], np.int32), np.array([1, 2], np.int32), 2, map_=[0, 0, 1])
    ctx.set_skin(2.0)
    r = ctx.rebomos_compute_host(2)
    assert np.isfinite(r["eng"]) and r["eng"] < 0
    ctx.close()


def _aeam_crowded():
    """one Si (angular) atom with 170 Al neighbours inside its 4.18 A cutoff: more than the 160 LDS slots"""
    pts = _sphere(170, 3.9)
    x = np.concatenate([[[0.0, 0.0, 0.0]], pts + np.array([0.0, 0.0, 0.0])]) + 30.0
    t = np.array([2] + [1] * 170, np.int32)
    return S.System(BOX, x, t, np.arange(1, 172, dtype=np.int32), np.array([0.0, 27.0, 28.0]))


@gpu
def test_aeam_angular_overflow_host_mode():
    s = _aeam_crowded()
    af = capi.AeamFile(POT_AEAM)
    tabs = af.build()
    ctx = capi.Context(0)
    ctx.aeam_set_tables(tabs)
    cut = af.cut_table(tabs) + 1.0
    nn, off, nb = S.neighbor_lists_cpu(s.x, s.type, s.n, cut)
    ctx.set_atoms_host(s.n, s.x, s.type, s.tag, 2)
    ctx.set_neighbors_csr_host(nn, off, nb, 1.0)
    code, msg = _code(ctx.aeam_density_host, s.n)
    assert code == MDP_EOVERFLOW and "Neighbor list overflow" in msg and "angular" in msg
    ctx.close()


@gpu
def test_overflow_on_a_force_only_step_is_sticky_until_the_next_host_read():
    """resident mode: force-only steps never read the flag words; the overflow must still stop the run at the
    next thermo instead of silently truncating forces (flags were cleared every compute)"""
    from lammps_plugins_amd.host import resident
    s = _aeam_crowded()
    af = capi.AeamFile(POT_AEAM)
    tabs = af.build()
    ctx = capi.Context(0)
    ctx.aeam_set_tables(tabs)
    s.mass[1:3] = af.mass[:2]
    cutghost = float(af.cut_table(tabs).max()) + 1.0
    dom = hostplan.Domain.single(ctx, capi.STYLE_AEAM, s, cutghost, 1.0, None)
    dom.build_neighbors()
    dom.compute(0, 0)          # overflow happens here, nobody looks
    dom.compute(0, 0)          # ... and the per-compute word is cleared here
    code, msg = _code(ctx.md_thermo)
    assert code == MDP_EOVERFLOW and "Neighbor list overflow" in msg
    ctx.close()


@gpu
def test_empty_subdomain_with_null_pointers():
    """a rank without atoms hands over NULL arrays (atom->x is NULL while nmax == 0): the reference loops over
    zero atoms; numpy's empty arrays have non-NULL data pointers, so call the C-ABI directly"""
    L = capi.lib()
    for style in ("rebomos", "aeam"):
        ctx = capi.Context(0)
        if style == "rebomos":
            ctx.rebomos_set_params(capi.read_rebomos_file(POT_REBOMOS))
        else:
            af = capi.AeamFile(POT_AEAM)
            ctx.aeam_set_tables(af.build())
        m = (C.c_int * 3)(0, 0, 1)
        assert L.mdp_set_atoms_host(ctx.h, 0, 0, None, None, None, 2, m) == 0
        eng = C.c_double(0.0)
        vir = (C.c_double * 6)()
        if style == "rebomos":
            assert L.mdp_set_skin(ctx.h, C.c_double(2.0)) == 0
            assert L.mdp_rebomos_check_host_list(ctx.h, 0, None, None, None, C.c_double(13.4)) == 0
            assert L.mdp_rebomos_compute_host(ctx.h, 3, 1, None, C.byref(eng), vir, None, None) == 0
            assert L.mdp_set_positions_host(ctx.h, None) == 0
            assert L.mdp_rebomos_compute_host(ctx.h, 0, 0, None, None, None, None, None) == 0
        else:
            assert L.mdp_set_neighbors_host(ctx.h, 0, 0, None, None, None, C.c_double(1.0)) == 0
            assert L.mdp_aeam_density_host(ctx.h, 3, None, None, C.byref(eng), None) == 0
            assert L.mdp_aeam_force_host(ctx.h, 3, 1, None, None, C.byref(eng), vir, None, None) == 0
        assert eng.value == 0.0 and list(vir) == [0.0] * 6
        # but NULL with a non-zero count is an argument error, not a crash
        assert L.mdp_set_atoms_host(ctx.h, 5, 0, None, None, None, 2, m) == MDP_EINVAL
        ctx.close()


@gpu
def test_host_list_guard_detects_exclusions_and_special_bits():
    """the reference walks the host's list entries (pair_rebomos.cpp:328-330); a host list with exclusions or
    special-bond bits must stop the run instead of being ignored"""
    import mdref
    import oracle_bindings as ob
    orc = ob.load()
    P = orc.rebomos_params(POT_REBOMOS)
    s = S.rebomos_bulk_cell()
    eng = mdref.RebomosCPU(orc, P, s)
    ctx = capi.Context(0)
    ctx.rebomos_set_params(capi.read_rebomos_file(POT_REBOMOS))
    xa = eng.all_positions(s.x)
    ctx.set_atoms_host(eng.nlocal, xa, eng.type_all, eng.tag_all, 2, map_=[0, 0, 1])
    ctx.set_skin(2.0)
    nall = len(xa)
    rows = [np.ascontiguousarray(eng.nb[eng.off[i]:eng.off[i] + eng.nn[i]], dtype=np.int32) for i in range(nall)]
    ilist = np.arange(eng.nlocal, dtype=np.int32)
    ctx.rebomos_check_host_list(ilist, eng.nn, rows, 13.4)            # the plain geometric list passes
    assert int(eng.nn[:eng.nlocal].sum()) == 142848                   # log.rebomos-bulk.1:82
    nn2 = eng.nn.copy()
    nn2[7] -= 1                                                        # one excluded pair
    code, msg = _code(ctx.rebomos_check_host_list, ilist, nn2, rows, 13.4)
    assert code == MDP_EINVAL and "not the plain geometric list" in msg
    rows2 = list(rows)
    rows2[0] = rows[0].copy()
    rows2[0][3] |= 1 << 30                                             # special-bond bits (SBBITS = 30)
    code, msg = _code(ctx.rebomos_check_host_list, ilist, eng.nn, rows2, 13.4)
    assert code == MDP_EINVAL and "special-bond bits" in msg
    ctx.close()


@gpu
def test_device_bytes_are_reported():
    ctx = capi.Context(0)
    before = ctx.device_bytes()
    ctx.rebomos_set_params(capi.read_rebomos_file(POT_REBOMOS))
    s = S.replicate(S.rebomos_bulk_cell(), (2, 2, 2))
    xa, ta, ga, owner, shift, nloc, ngh = S.with_ghosts(s, 13.4)
    ctx.set_atoms_host(nloc, xa, ta, ga, 2, map_=[0, 0, 1])
    ctx.set_skin(2.0)
    ctx.rebomos_compute_host(nloc)
    assert ctx.device_bytes() > before + 24.0 * len(xa)                # at least the coordinates
    ctx.close()


@gpu
@pytest.mark.parametrize("register", ["0", "1"])
def test_host_reallocates_x_between_steps(register, monkeypatch):
    """LAMMPS re-allocates atom->x when nmax grows (memory->grow at migration): every step here hands over a NEW
    array -- sometimes at the address of one just freed -- in the default staged mode and with in-place
    registration (MDP_HOST_REGISTER=1, arrays >= 8 MB, mdp_host_release before the free).  Forces must follow the
    positions actually handed over (stale registrations keyed by pointer)."""
    monkeypatch.setenv("MDP_HOST_REGISTER", register)
    ctx = capi.Context(0)
    ctx.rebomos_set_params(capi.read_rebomos_file(POT_REBOMOS))
    s = S.replicate(S.rebomos_bulk_cell(), (11, 11, 11))            # 383 k atoms + ghosts: x is > 8 MB
    xa, ta, ga, owner, shift, nloc, ngh = S.with_ghosts(s, 13.4)
    ctx.set_atoms_host(nloc, xa, ta, ga, 2, map_=[0, 0, 1])
    ctx.set_skin(2.0)
    f0 = ctx.rebomos_compute_host(nloc, eflag=0, vflag=0)["f"]
    L = capi.lib()
    rng = np.random.default_rng(1)
    disp = np.zeros_like(xa)
    for step in range(3):
        d_owned = 0.02 * rng.standard_normal((nloc, 3))
        disp[:nloc] += d_owned
        disp[nloc:] = disp[owner]
        xnew = np.ascontiguousarray(xa + disp)                     # a fresh allocation every step
        ctx.set_positions_host(xnew)
        f = ctx.rebomos_compute_host(nloc, eflag=0, vflag=0)["f"]
        if register == "1":
            assert L.mdp_host_release(ctx.h, capi.C.c_void_p(xnew.ctypes.data)) == 0
        del xnew
    # reference: a fresh context given the final positions at once
    ctx2 = capi.Context(0)
    ctx2.rebomos_set_params(capi.read_rebomos_file(POT_REBOMOS))
    ctx2.set_atoms_host(nloc, xa + disp, ta, ga, 2, map_=[0, 0, 1])
    ctx2.set_skin(2.0)
    fref = ctx2.rebomos_compute_host(nloc, eflag=0, vflag=0)["f"]
    assert np.abs(f - fref).max() < 1e-9
    assert np.abs(f - f0).max() > 1e-3                             # and they really changed
    ctx.close()
    ctx2.close()


@gpu
def test_candidate_rows_that_overflow_only_because_of_the_skin_are_rebuilt_with_a_smaller_skin(oracle):
    """a simple-cubic block of S atoms 1.5 A apart: 79 atoms within rcmax + 1 A of an interior atom (more than the
    64 bits of the active mask) but 33 inside rcmax itself, which the reference would simply list
    (pair_rebomos.cpp:337-350, oneatom = 2000).  The style halves its inner skin until the rows fit and computes;
    forces equal the oracle's (interior centres go through the general kernel: more than 32 neighbours)."""
    import mdref
    g = np.arange(7) * 1.5
    x = np.array([[a, b, c] for a in g for b in g for c in g]) + 25.0
    n = len(x)
    s = S.System(BOX, x, np.full(n, 2, np.int32), np.arange(1, n + 1, dtype=np.int32), np.array([0.0, 95.95, 32.065]))
    ctx = capi.Context(0)
    ctx.rebomos_set_params(capi.read_rebomos_file(POT_REBOMOS))
    ctx.set_atoms_host(n, s.x, s.type, s.tag, 2, map_=[0, 0, 1])
    ctx.set_skin(2.0)
    r = ctx.rebomos_compute_host(n)
    P = oracle.rebomos_params(POT_REBOMOS)
    o = mdref.RebomosCPU(oracle, P, s).compute(s.x)
    scale = np.abs(o["f_owned"]).max()
    assert np.abs(r["f"] - o["f_owned"]).max() < 1e-11 * scale
    assert r["eng"] == pytest.approx(o["eng"], rel=1e-11)
    ctx.close()
