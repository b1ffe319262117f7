"""Resident-mode MD driver (host side): builds a sub-domain (owned atoms + ghost map), uploads it
once, then steps entirely on the GPU through the C-ABI.  This is the small slice of the LAMMPS host
around Pair::compute() that the bench and the tests need (Verlet::run loop of fix nve, thermo,
`neigh_modify every 1 check yes`); the arithmetic all happens in libmdpair_hip.so."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from . import system as S


def morton_order(x: np.ndarray, lo: np.ndarray, cell: float) -> np.ndarray:
    """argsort of atoms along a Z-order curve on a `cell`-sized grid (spatial locality for gathers)"""
    g = np.floor((x - lo) / cell).astype(np.int64)
    g -= g.min(axis=0)
    g = np.minimum(g, (1 << 20) - 1).astype(np.uint64)

    def spread(v):
        v = v & np.uint64(0x1FFFFF)
        v = (v | (v << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
        v = (v | (v << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
        v = (v | (v << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
        return v

    key = spread(g[:, 0]) | (spread(g[:, 1]) << np.uint64(1)) | (spread(g[:, 2]) << np.uint64(2))
    return np.argsort(key, kind="stable")


class Domain:
    """One GPU sub-domain in resident mode.

    owned atoms: x,v,type,tag (already restricted to this sub-domain)
    ghosts: owner (local index or -1), shift (Cartesian image shift, or absolute position if owner<0)
    """

    def __init__(self, ctx: capi.Context, style: int, box: S.Box, x, v, type_, tag, mass, map_, ghost_owner,
                 ghost_shift, ghost_type, ghost_tag, skin: float, dt: float = 0.001):
        self.ctx, self.style, self.box = ctx, style, box
        self.nlocal, self.nghost = len(x), len(ghost_owner)
        self.mass = np.asarray(mass, dtype=np.float64)
        self.skin, self.dt = skin, dt
        xg = np.where((ghost_owner >= 0)[:, None], x[np.maximum(ghost_owner, 0)], 0.0) + ghost_shift \
            if self.nghost else np.zeros((0, 3))
        allx = np.concatenate([x, xg]) if self.nghost else x
        pad = 1.0 + skin
        cfg = capi.MdConfig()
        cfg.style, cfg.nlocal, cfg.nghost, cfg.ntypes = style, self.nlocal, self.nghost, len(mass) - 1
        cfg.skin, cfg.dt, cfg.ftm2v, cfg.mvv2e = skin, dt, S.FTM2V, S.MVV2E
        lo, hi = allx.min(axis=0) - pad, allx.max(axis=0) + pad
        for d in range(3):
            cfg.bbox_lo[d], cfg.bbox_hi[d] = lo[d], hi[d]
        self.cfg = cfg
        ctx.md_setup(cfg, x, v, type_, tag, mass, map_, ghost_owner, ghost_shift, ghost_type, ghost_tag)
        self.natoms_total = self.nlocal  # overwritten by the multi-rank driver
        self.builds = 0

    @classmethod
    def single(cls, ctx, style, s: S.System, cutghost: float, skin: float, map_, v0=None, dt=0.001, sort=True):
        """whole periodic box on one GPU: ghosts are periodic self-images"""
        x = S.wrap(s.box, s.x)
        v = np.zeros_like(x) if v0 is None else np.asarray(v0, dtype=np.float64)
        t, g = s.type, s.tag
        if sort:
            order = morton_order(x, s.box.lo, 3.0)
            x, v, t, g = x[order], v[order], t[order], g[order]
        owner, shift = S.make_ghosts(s.box, x, cutghost)
        shift_cart = shift @ s.box.h.T
        if sort and len(owner):
            go = morton_order(x[owner] + shift_cart, s.box.lo - cutghost - 1.0, 3.0)
            owner, shift_cart = owner[go], shift_cart[go]
        d = cls(ctx, style, s.box, np.ascontiguousarray(x), np.ascontiguousarray(v), t, g, s.mass, map_,
                owner.astype(np.int32), np.ascontiguousarray(shift_cart), t[owner], g[owner], skin, dt)
        d.order_tag = g
        return d

    # ------------------------------------------------------------------ MD
    def build_neighbors(self):
        self.ctx.md_build_neighbors()
        self.builds += 1

    def compute(self, eflag=0, vflag=0):
        self.ctx.md_compute(eflag, vflag)

    def step(self, eflag=0, vflag=0):
        """one velocity-Verlet step (Verlet::run body): initial_integrate, [neighbor], force, final"""
        self.ctx.md_initial_integrate()
        self.ctx.md_compute(eflag, vflag)
        self.ctx.md_final_integrate()

    def thermo(self, natoms_total=None, volume=None):
        t = self.ctx.md_thermo()
        n = self.natoms_total if natoms_total is None else natoms_total
        vol = self.box.volume if volume is None else volume
        t["temp"] = S.temperature(t["ke"], n)
        t["press"] = S.pressure(t["ke"], t["virial"], n, vol)
        return t

    def needs_rebuild(self, thermo=None) -> bool:
        """`neigh_modify check yes`: any atom moved more than skin/2 since the last build"""
        t = self.ctx.md_thermo() if thermo is None else thermo
        return t["maxdisp2"] > (0.5 * self.skin) ** 2
