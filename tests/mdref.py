"""Small CPU MD driver around the oracle (tests only): velocity-Verlet NVE + thermo per
SURVEY.md Appendix B.  Used to pin the oracle to USER-REBOMOS/log.rebomos-bulk.1."""
from __future__ import annotations

import numpy as np

import oracle_bindings as ob
from lammps_plugins_amd.host import system as S


class RebomosCPU:
    """REBO-MoS on one periodic domain through the oracle (ghost images + list built once)."""

    def __init__(self, orc, P, s: S.System, skin: float = 2.0):
        self.orc, self.P, self.s = orc, P, s
        self.cutneigh = P.cut3rebo + skin
        (self.x_all, self.type_all, self.tag_all, self.owner, self.shift, self.nlocal,
         self.nghost) = S.with_ghosts(s, self.cutneigh)
        self.elem = (self.type_all - 1).astype(np.int32)
        rcmax = np.array([[P.rcmax[a][b] for b in range(2)] for a in range(2)])
        cg = np.zeros((3, 3))
        cg[1:, 1:] = rcmax + skin
        self.nn, self.off, self.nb = S.neighbor_lists_cpu(self.x_all, self.type_all, self.nlocal, self.cutneigh, cg)

    def all_positions(self, x):
        return np.concatenate([x, x[self.owner] + self.shift])

    def compute(self, x, eflag=3, vflag=5, phases=3):
        o = self.orc.rebomos_compute(self.P, self.nlocal, self.all_positions(x), self.elem, self.tag_all, self.nn,
                                     self.off, self.nb, eflag=eflag, vflag=vflag, phases=phases)
        o["f_owned"] = ob.fold_ghost_forces(o["f"], self.owner, self.nlocal)
        o["eatom_owned"] = ob.fold_ghost_forces(o["eatom"][:, None], self.owner, self.nlocal)[:, 0]
        return o


class AeamCPU:
    def __init__(self, orc, T, s: S.System, skin: float = 1.0):
        self.orc, self.T, self.s = orc, T, s
        ne = T.nelements
        cut = np.zeros((ne + 1, ne + 1))
        for a in range(ne):
            for b in range(ne):
                cut[a + 1, b + 1] = T.cut[a][b]
        self.cut = cut
        self.cutmax = float(cut.max())
        (self.x_all, self.type_all, self.tag_all, self.owner, self.shift, self.nlocal,
         self.nghost) = S.with_ghosts(s, self.cutmax + skin)
        self.nn, self.off, self.nb = S.neighbor_lists_cpu(self.x_all, self.type_all, self.nlocal, cut + skin)

    def all_positions(self, x):
        return np.concatenate([x, x[self.owner] + self.shift])

    def compute(self, x, eflag=3, vflag=5):
        o = self.orc.aeam_compute(self.T, self.nlocal, self.all_positions(x), self.type_all, self.nn, self.off,
                                  self.nb, eflag=eflag, vflag=vflag)
        o["f_owned"] = ob.fold_ghost_forces(o["f"], self.owner, self.nlocal)
        return o


def nve(engine, s: S.System, nsteps: int, dt: float = 0.001, thermo_every: int = 10, v0=None):
    """returns list of thermo rows dict(step,temp,press,pe,ke)"""
    m = s.mass[s.type]
    x = s.x.copy()
    v = np.zeros_like(x) if v0 is None else v0.copy()
    rows = []

    def thermo(step, o):
        ke = S.kinetic_energy(m, v)
        rows.append(dict(step=step, temp=S.temperature(ke, s.n), pe=o["eng"], ke=ke,
                         press=S.pressure(ke, o["virial_fdotr"], s.n, s.box.volume)))

    o = engine.compute(x)
    f = o["f_owned"]
    thermo(0, o)
    for step in range(1, nsteps + 1):
        v += 0.5 * dt * S.FTM2V * f / m[:, None]
        x += dt * v
        o = engine.compute(x)
        f = o["f_owned"]
        v += 0.5 * dt * S.FTM2V * f / m[:, None]
        if step % thermo_every == 0:
            thermo(step, o)
    return rows, x, v
