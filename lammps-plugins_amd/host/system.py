"""Host-side system construction: the part of the LAMMPS host that sits *before*
the pair-style hot path (lattice/create_atoms, replicate, periodic ghost images,
CPU neighbor lists for small test cells).

Semantics follow SURVEY.md Appendix B (the recipe that reproduces
USER-REBOMOS/log.rebomos-bulk.1) and USER-REBOMOS/in.rebomos-bulk:3-25,
USER-AEAM/sample.in:6-19.  Pure numpy/scipy; nothing here touches the GPU or the
oracle.
"""
from __future__ import annotations

import dataclasses
import numpy as np

# LAMMPS "metal" unit constants
BOLTZ = 8.617343e-5
MVV2E = 1.0364269e-4
FTM2V = 1.0 / 1.0364269e-4
NKTV2P = 1.6021765e6
NEIGHMASK = 0x1FFFFFFF


def mul_upper(a: np.ndarray, m: np.ndarray) -> np.ndarray:
    """rows of `a` times the transpose of the upper-triangular 3x3 matrix m (= a @ m.T), element by element"""
    a = np.asarray(a, dtype=np.float64)
    out = np.empty_like(a)
    out[..., 0] = m[0, 0] * a[..., 0] + m[0, 1] * a[..., 1] + m[0, 2] * a[..., 2]
    out[..., 1] = m[1, 1] * a[..., 1] + m[1, 2] * a[..., 2]
    out[..., 2] = m[2, 2] * a[..., 2]
    return out


@dataclasses.dataclass
class Box:
    """LAMMPS (restricted-)triclinic box: edge vectors a=(xprd,0,0), b=(xy,yprd,0), c=(xz,yz,zprd)."""
    lo: np.ndarray          # (3,)
    prd: np.ndarray         # (3,) xprd yprd zprd
    tilt: np.ndarray        # (3,) xy xz yz

    @property
    def h(self) -> np.ndarray:
        xy, xz, yz = self.tilt
        return np.array([[self.prd[0], xy, xz], [0.0, self.prd[1], yz], [0.0, 0.0, self.prd[2]]])

    @property
    def hinv(self) -> np.ndarray:
        # Domain::set_global_box: closed-form inverse of the upper-triangular h (no LAPACK call)
        xprd, yprd, zprd = self.prd
        xy, xz, yz = self.tilt
        return np.array([[1.0 / xprd, -xy / (xprd * yprd), (yz * xy - yprd * xz) / (xprd * yprd * zprd)],
                         [0.0, 1.0 / yprd, -yz / (yprd * zprd)],
                         [0.0, 0.0, 1.0 / zprd]])

    @property
    def volume(self) -> float:
        return float(np.prod(self.prd))

    # Domain::x2lamda / lamda2x written out term by term (as LAMMPS and csrc/domain.hip do).  No `@`: numpy hands a
    # matrix product to the BLAS library, and concurrent products from several Python threads -- the rank threads of
    # resident.run_ranks -- returned wrong rows now and then (measured here: `a @ b.T` in 8 threads), which put atoms
    # into the wrong brick or on top of each other before the device library ever saw them.
    def x2lamda(self, x: np.ndarray) -> np.ndarray:
        return mul_upper(np.asarray(x, dtype=np.float64) - self.lo, self.hinv)

    def lamda2x(self, lam: np.ndarray) -> np.ndarray:
        return mul_upper(np.asarray(lam, dtype=np.float64), self.h) + self.lo

    def replicate(self, n) -> "Box":
        n = np.asarray(n, dtype=float)
        xy, xz, yz = self.tilt
        return Box(self.lo.copy(), self.prd * n, np.array([xy * n[1], xz * n[2], yz * n[2]]))


@dataclasses.dataclass
class System:
    box: Box
    x: np.ndarray        # (n,3) float64, owned atoms
    type: np.ndarray     # (n,) int32, 1-based LAMMPS types
    tag: np.ndarray      # (n,) int32, 1-based atom IDs
    mass: np.ndarray     # (ntypes+1,) per-type masses, index 0 unused

    @property
    def n(self) -> int:
        return int(self.x.shape[0])


# ----------------------------------------------------------------------------------------
# lattices
# ----------------------------------------------------------------------------------------

def rebomos_bulk_cell() -> System:
    """The 288-atom triclinic 2H-MoS2 cell of USER-REBOMOS/in.rebomos-bulk:3-25
    (lattice custom ... origin 0.1; region prism 0 4 0 8 0 1 tilt -2 0 0)."""
    a1 = np.array([3.1903157234, 0.0, 0.0])
    a2 = np.array([-1.5964590311, 2.7651481541, 0.0])
    a3 = np.array([0.0, 0.0, 13.9827680588])
    basis = np.array([
        [0.0, 0.0, 3.0 / 4.0], [0.0, 0.0, 1.0 / 4.0],
        [2.0 / 3.0, 1.0 / 3.0, 0.862008989], [1.0 / 3.0, 2.0 / 3.0, 0.137990996],
        [1.0 / 3.0, 2.0 / 3.0, 0.362008989], [2.0 / 3.0, 1.0 / 3.0, 0.637991011]])
    btype = np.array([1, 1, 2, 2, 2, 2], dtype=np.int32)
    # lattice spacings = extent of the unit cell's bounding box (log.rebomos-bulk.1:17)
    corners = np.array([i * a1 + j * a2 + k * a3 for i in (0, 1) for j in (0, 1) for k in (0, 1)])
    lat = corners.max(axis=0) - corners.min(axis=0)
    origin = 0.1 * lat
    box = Box(np.zeros(3), np.array([4 * lat[0], 8 * lat[1], 1 * lat[2]]), np.array([-2.0 * lat[0], 0.0, 0.0]))
    xs, ts = [], []
    # creation order: k outer, j, i, basis inner (tags follow it)
    for k in range(-2, 3):
        for j in range(-3, 12):
            for i in range(-10, 14):
                for b in range(6):
                    p = (i + basis[b, 0]) * a1 + (j + basis[b, 1]) * a2 + (k + basis[b, 2]) * a3 + origin
                    xs.append(p)
                    ts.append(btype[b])
    xs = np.array(xs)
    ts = np.array(ts, dtype=np.int32)
    lam = box.x2lamda(xs)
    keep = np.all((lam >= -1.0e-6) & (lam < 1.0 - 2.0e-6), axis=1)
    xs, ts = xs[keep], ts[keep]
    mass = np.array([0.0, 95.95, 32.065])
    return System(box, np.ascontiguousarray(xs), ts, np.arange(1, len(xs) + 1, dtype=np.int32), mass)


def fcc_cell(a: float, n, frac_type2: float = 0.0, seed: int = 7683797,
             mass=(0.0, 27.0, 28.0)) -> System:
    """fcc lattice of n=(nx,ny,nz) conventional cells (USER-AEAM/sample.in:8-11); a fraction of atoms
    is switched to type 2 by a seeded numpy RNG (LAMMPS' own RNG for `set type/fraction` is out of scope)."""
    nx, ny, nz = (n, n, n) if np.isscalar(n) else n
    basis = np.array([[0, 0, 0], [0.5, 0.5, 0], [0.5, 0, 0.5], [0, 0.5, 0.5]])
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    cells = np.stack([i.ravel(), j.ravel(), k.ravel()], axis=1).astype(float)
    x = (cells[:, None, :] + basis[None, :, :]).reshape(-1, 3) * a
    t = np.ones(len(x), dtype=np.int32)
    if frac_type2 > 0:
        rng = np.random.default_rng(seed)
        t[rng.random(len(x)) < frac_type2] = 2
    box = Box(np.zeros(3), np.array([nx * a, ny * a, nz * a]), np.zeros(3))
    return System(box, np.ascontiguousarray(x), t, np.arange(1, len(x) + 1, dtype=np.int32), np.array(mass, dtype=float))


def replicate(s: System, n) -> System:
    """LAMMPS `replicate nx ny nz`: copies shifted by whole box vectors."""
    nx, ny, nz = n
    h = s.box.h
    xs, ts = [], []
    for k in range(nz):
        for j in range(ny):
            for i in range(nx):
                xs.append(s.x + i * h[:, 0] + j * h[:, 1] + k * h[:, 2])
                ts.append(s.type)
    x = np.ascontiguousarray(np.concatenate(xs))
    return System(s.box.replicate(n), x, np.concatenate(ts), np.arange(1, len(x) + 1, dtype=np.int32), s.mass)


def wrap(box: Box, x: np.ndarray) -> np.ndarray:
    """remap positions into the periodic box (LAMMPS Domain::remap at reneighboring)"""
    lam = box.x2lamda(x)
    return np.ascontiguousarray(box.lamda2x(lam - np.floor(lam)))


def jitter(s: System, amp: float, seed: int) -> System:
    rng = np.random.default_rng(seed)
    return dataclasses.replace(s, x=wrap(s.box, s.x + rng.uniform(-amp, amp, s.x.shape)))


def scale(s: System, fac: float) -> System:
    b = s.box
    return dataclasses.replace(s, box=Box(b.lo * fac, b.prd * fac, b.tilt * fac), x=s.x * fac)


# ----------------------------------------------------------------------------------------
# ghosts (periodic images), SURVEY.md Appendix B "Ghosts"
# ----------------------------------------------------------------------------------------

def ghost_cut_lamda(box: Box, cut: float) -> np.ndarray:
    """ghost-shell half width in lamda units: cut * |row_d(h^-1)|"""
    return cut * np.linalg.norm(box.hinv, axis=1)


def make_ghosts(box: Box, x: np.ndarray, cut: float, sublo=None, subhi=None, lam: np.ndarray | None = None,
                owned_mask: np.ndarray | None = None):
    """All periodic images (of atoms anywhere in the box) that fall in the ghost shell of the
    sub-domain [sublo, subhi) (lamda units; default: whole box), excluding the un-shifted image of
    the atoms the sub-domain owns (owned_mask; default: atoms whose lamda lies in [sublo,subhi),
    or every atom for the whole box -- atoms may sit slightly outside the box between reneighborings).
    Returns (owner_index, shift) with ghost position = x[owner] + shift @ h^T."""
    whole = sublo is None and subhi is None
    sublo = np.zeros(3) if sublo is None else np.asarray(sublo, float)
    subhi = np.ones(3) if subhi is None else np.asarray(subhi, float)
    lam = box.x2lamda(x) if lam is None else lam
    if owned_mask is None:
        owned_mask = np.ones(len(x), bool) if whole else np.all((lam >= sublo) & (lam < subhi), axis=1)
    c = ghost_cut_lamda(box, cut)
    lo, hi = sublo - c, subhi + c
    owners, shifts = [], []
    rng = [np.arange(int(np.floor(lo[d] - 1.0)), int(np.ceil(hi[d])) + 1) for d in range(3)]
    # per-dimension masks once per shift value, then AND the three: 27 cheap passes for a big box
    masks = [{int(sv): (lam[:, d] + sv >= lo[d]) & (lam[:, d] + sv < hi[d]) for sv in rng[d]} for d in range(3)]
    masks = [{k: m for k, m in md.items() if m.any()} for md in masks]
    for sx, mx in masks[0].items():
        for sy, my in masks[1].items():
            mxy = mx & my
            if not mxy.any():
                continue
            for sz, mz in masks[2].items():
                inshell = mxy & mz
                if sx == 0 and sy == 0 and sz == 0:
                    inshell = inshell & ~owned_mask
                idx = np.nonzero(inshell)[0]
                if len(idx):
                    owners.append(idx)
                    shifts.append(np.broadcast_to(np.array([sx, sy, sz], dtype=float), (len(idx), 3)))
    if not owners:
        return np.zeros(0, dtype=np.int64), np.zeros((0, 3))
    return np.concatenate(owners), np.concatenate(shifts)


def with_ghosts(s: System, cut: float):
    """Single-domain system: returns (x_all, type_all, tag_all, owner, shift_cart, nlocal, nghost)."""
    owner, shift = make_ghosts(s.box, s.x, cut)
    shift_cart = mul_upper(shift, s.box.h)
    xg = s.x[owner] + shift_cart
    x_all = np.ascontiguousarray(np.concatenate([s.x, xg]))
    type_all = np.concatenate([s.type, s.type[owner]]).astype(np.int32)
    tag_all = np.concatenate([s.tag, s.tag[owner]]).astype(np.int32)
    return x_all, type_all, tag_all, owner.astype(np.int32), shift_cart, s.n, len(owner)


# ----------------------------------------------------------------------------------------
# CPU neighbor lists for small cells (tests / oracle side).  The product builds its lists
# on the device (csrc/neighbor.hip) or receives them from the LAMMPS host.
# ----------------------------------------------------------------------------------------

def neighbor_lists_cpu(x_all: np.ndarray, type_all: np.ndarray, nlocal: int, cutneigh_owned,
                       cutneigh_ghost=None):
    """Full neighbor lists in CSR form.

    owned i:  all j != i (owned or ghost) with rsq <= cutneigh_owned[ti][tj]^2
              (a scalar or an (ntypes+1,ntypes+1) table: LAMMPS builds with per-type-pair cutoffs)
    ghost i:  (if cutneigh_ghost is given, REQ_GHOST) all j with rsq <= cutneigh_ghost[ti][tj]^2
    Returns numneigh (nall,) int32, offset (nall+1,) int64, neigh (total,) int32.
    """
    from scipy.spatial import cKDTree
    nall = len(x_all)
    tree = cKDTree(x_all)
    ntypes = int(type_all.max())

    def table(c):
        c = np.asarray(c, dtype=float)
        if c.ndim == 0:
            c = np.full((ntypes + 1, ntypes + 1), float(c))
        return c

    def build(idx, cut):
        cut = table(cut)
        rmax = float(cut.max())
        lists = tree.query_ball_point(x_all[idx], rmax * (1.0 + 1e-9) + 1e-9, return_sorted=True)
        out = []
        for i, js in zip(idx, lists):
            js = np.asarray(js, dtype=np.int64)
            js = js[js != i]
            d = x_all[i] - x_all[js]
            rsq = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
            c = cut[type_all[i], type_all[js]]
            out.append(js[rsq <= c * c].astype(np.int32))
        return out

    lists = build(np.arange(nlocal), cutneigh_owned)
    if cutneigh_ghost is not None and nall > nlocal:
        lists += build(np.arange(nlocal, nall), cutneigh_ghost)
    else:
        lists += [np.zeros(0, dtype=np.int32)] * (nall - nlocal)
    numneigh = np.array([len(l) for l in lists], dtype=np.int32)
    offset = np.zeros(nall + 1, dtype=np.int64)
    np.cumsum(numneigh, out=offset[1:])
    neigh = np.concatenate(lists).astype(np.int32) if offset[-1] else np.zeros(0, dtype=np.int32)
    return numneigh, offset, neigh


# ----------------------------------------------------------------------------------------
# thermo
# ----------------------------------------------------------------------------------------

def kinetic_energy(mass_per_atom: np.ndarray, v: np.ndarray) -> float:
    return 0.5 * MVV2E * float(np.sum(mass_per_atom * np.sum(v * v, axis=1)))


def temperature(ke: float, natoms: int) -> float:
    dof = 3 * natoms - 3
    return 2.0 * ke / (dof * BOLTZ)


def pressure(ke: float, virial6, natoms: int, volume: float) -> float:
    dof = 3 * natoms - 3
    t = temperature(ke, natoms)
    return (dof * BOLTZ * t + virial6[0] + virial6[1] + virial6[2]) / (3.0 * volume) * NKTV2P


def gaussian_velocities(s: System, temp: float, seed: int) -> np.ndarray:
    """`velocity all create T seed`-like: Gaussian, zero net momentum, rescaled to exactly T."""
    rng = np.random.default_rng(seed)
    m = s.mass[s.type]
    v = rng.standard_normal(s.x.shape) / np.sqrt(m)[:, None]
    v -= (m[:, None] * v).sum(axis=0) / m.sum()
    t = temperature(kinetic_energy(m, v), s.n)
    return v * np.sqrt(temp / t)
