"""The fixture set of SURVEY.md Appendix C: constructions, the oracle run that freezes a case, and the (de)serialisation
of a case to a small .npz under tests/golden/fixtures/.

WHAT THESE FIXTURES PIN -- AND WHAT THEY DO NOT.  The outputs stored in the .npz files come from THIS repository's CPU
oracle (oracle/*.c), not from the reference: the reference cannot be built in this image (it needs LAMMPS headers,
DESIGN.md section 5).  They add no new tie to the reference beyond what tests/test_oracle_rebomos.py already has
(log.rebomos-bulk.1:54-56, which `R-bulk-0` reproduces) -- AEAM stays "parity unpinned".  What they do is freeze the
oracle: a later edit that changes oracle and kernel alike (the one failure a live-oracle parity test cannot see) now
fails tests/test_golden_fixtures.py on the CPU, and the GPU parity tests of tests/test_gpu_golden.py compare the HIP path
with the frozen numbers without calling the oracle at all.

A fixture = inputs {box, x, type, tag, mass} + outputs {f folded onto the owners and the raw owned+ghost f, eng_vdwl,
virial[6] (fdotr), the tallied virial, eatom and vatom folded onto the owners; REBO: nM, nS, REBO_numneigh; AEAM: rho,
fp}.  Ghosts and neighbor lists are not stored: they are rebuilt from the inputs by the same deterministic host code
(lammps_plugins_amd.host.system: images in a fixed order, cKDTree lists sorted by index), so the summation order of the
oracle -- which follows the list order (pair_rebomos.cpp:394-402, pair_aeam.cpp:174-252) -- is reproduced too."""
from __future__ import annotations

import dataclasses
import os

import numpy as np

from lammps_plugins_amd.host import system as S
import mdref

HERE = os.path.dirname(os.path.abspath(__file__))
FIXDIR = os.path.join(HERE, "golden", "fixtures")

AEAM_MASS = (0.0, 26.98, 28.086)      # AlSi.aeam:13-14 (only the integrator reads masses; kept with the inputs)


def _aeam_cell(ncell, frac, amp, seed):
    s = S.fcc_cell(4.045, ncell, frac_type2=frac, seed=seed, mass=AEAM_MASS)
    return S.jitter(s, amp, seed=seed + 1) if amp else s


def _cluster(x, types, mass, box=60.0):
    x = np.asarray(x, dtype=float).reshape(-1, 3) + 0.5 * box
    return S.System(S.Box(np.zeros(3), np.full(3, box), np.zeros(3)), x, np.asarray(types, dtype=np.int32),
                    np.arange(1, len(x) + 1, dtype=np.int32), np.array(mass))


def _aeam_edge():
    """A-edge: hand-built clusters, one periodic box, 25 A apart (beyond every cutoff + skin), so that each cluster sees
    only itself: isolated Si (rho = 0 -> Fptmp = 0, pair_aeam.cpp:329-332; row clamp m = 1, :286), isolated Al, Si with
    one neighbour (no triplet), pairs just below / above each cut and cut - 1.5 (:187-194, 350, 408-418), the last table
    row with p = min(p, 1) (:198-201), a dense pair (upper rows of the embedding table)."""
    clusters = [
        ([[0, 0, 0]], [2]), ([[0, 0, 0]], [1]),
        ([[0, 0, 0], [2.6, 0, 0]], [2, 1]), ([[0, 0, 0], [2.35, 0, 0]], [2, 2]),
        ([[0, 0, 0], [6.5 - 1e-9, 0, 0]], [1, 1]), ([[0, 0, 0], [6.5 + 1e-9, 0, 0]], [1, 1]),
        ([[0, 0, 0], [4.18 - 1e-9, 0, 0], [0, 4.18 + 1e-9, 0]], [2, 1, 1]),
        ([[0, 0, 0], [4.5, 0, 0], [0, 2.5, 0]], [2, 2, 1]),
        ([[0, 0, 0], [5.28 - 1.5 - 1e-9, 0, 0], [0, 5.28 - 1.5 + 1e-9, 0], [0, 0, 2.4]], [2, 2, 2, 1]),
        ([[0, 0, 0], [2.5, 0, 0], [-1.2, 2.2, 0], [-1.2, -2.2, 0.3]], [2, 1, 1, 1]),
        ([[0, 0, 0], [2.4, 0, 0], [-1.2, 2.1, 0], [0, -1, 2.2], [0.5, 0.5, -2.4]], [2, 2, 1, 2, 1]),
        ([[0, 0, 0], [1.9, 0, 0]], [1, 1]),
    ]
    xs, ts = [], []
    for k, (x, t) in enumerate(clusters):
        origin = np.array([(k % 3) * 25.0, ((k // 3) % 3) * 25.0, (k // 9) * 25.0]) - 30.0
        xs.append(np.asarray(x, dtype=float) + origin)
        ts += t
    return _cluster(np.vstack(xs), ts, AEAM_MASS, box=90.0)


def _rebomos(fac, amp, seed, rep=None):
    s = S.rebomos_bulk_cell()
    if rep:
        s = S.replicate(s, rep)
    if fac != 1.0:
        s = S.scale(s, fac)
    return S.jitter(s, amp, seed=seed) if amp else s


# name -> (style, constructor).  R-bulk-0 is special: three states of the in.rebomos-bulk run (steps 0, 10, 20).
CASES = {
    "R-strain-112": ("rebomos", lambda: _rebomos(1.12, 0.15, 1234)),   # all branches: switching interior, LJ cubic
    "R-comp-093": ("rebomos", lambda: _rebomos(0.93, 0.10, 77)),       # denser REBO lists
    "R-jit-100": ("rebomos", lambda: _rebomos(1.0, 0.15, 5)),          # near-equilibrium noise
    "R-repl-2": ("rebomos", lambda: _rebomos(1.0, 0.0, 0, rep=(2, 2, 2))),   # replicate KAT: PE = 8 x
    "A-6-8pct": ("aeam", lambda: _aeam_cell(6, 0.08, 0.075, 99)),      # every pair type, mixed angular triplets
    "A-20-sample": ("aeam", lambda: _aeam_cell(20, 0.0075, 0.0, 7683797)),       # sample.in-sized, perfect lattice
    "A-20-sample-jit": ("aeam", lambda: _aeam_cell(20, 0.0075, 0.05, 7683797)),  # ... with thermal-like noise
    "A-edge": ("aeam", _aeam_edge),
}
BULK_STEPS = (0, 10, 20)
# the 32 000-atom cases keep per-atom outputs for every SAMPLE_EVERY-th atom (plus all angular atoms and their sums)
SAMPLE_EVERY = 16
LARGE = 8000


def engine(style, s, orc, P=None, T=None):
    return mdref.RebomosCPU(orc, P, s) if style == "rebomos" else mdref.AeamCPU(orc, T, s)


class _RebomosCuts:
    """what mdref.RebomosCPU reads of the parameter struct to build ghosts and lists"""

    def __init__(self, rcmax):
        self.rcmax = rcmax
        self.cut3rebo = 3.0 * rcmax[0][0]              # pair_rebomos.cpp:257


class _AeamCuts:
    def __init__(self, cut):                           # cut[a][b], elements 0-based (setfl->cut)
        self.nelements, self.cut = len(cut), cut


def lists_only(style, s, cuts):
    """ghosts + neighbor lists of a fixture WITHOUT the oracle (the GPU tests): `cuts` = rcmax[2][2] of the product's
    own parameter file parser (rebomos) or its cut[ne][ne] table (aeam).  compute() of the result must not be called."""
    return mdref.RebomosCPU(None, _RebomosCuts(cuts), s) if style == "rebomos" else mdref.AeamCPU(None, _AeamCuts(cuts), s)


def _fold(a, owner, nlocal):
    out = a[:nlocal].copy()
    np.add.at(out, owner, a[nlocal:])
    return out


def oracle_outputs(style, eng, x):
    """everything a fixture stores, from one oracle compute() with every tally on"""
    o = eng.compute(x, eflag=3, vflag=5)
    n = eng.nlocal
    out = dict(f=_fold(o["f"], eng.owner, n), f_raw=o["f"], eng=np.float64(o["eng"]),
               virial_fdotr=o["virial_fdotr"], virial_tally=o["virial_tally"],
               eatom=_fold(o["eatom"], eng.owner, n), vatom=_fold(o["vatom"], eng.owner, n))
    if style == "rebomos":
        out.update(nM=o["nM"][:n], nS=o["nS"][:n], rebo_numneigh=o["rebo_numneigh"][:n])
    else:
        out.update(rho=o["rho"][:n], fp=o["fp"][:n])
    return out


PER_ATOM = ("f", "eatom", "vatom", "nM", "nS", "rebo_numneigh", "rho", "fp")


def sample_index(s):
    if s.n < LARGE:
        return None
    keep = np.zeros(s.n, dtype=bool)
    keep[::SAMPLE_EVERY] = True
    keep[s.type != 1] = True
    return np.nonzero(keep)[0]


def pack(style, s, out):
    """dict of arrays for np.savez_compressed"""
    d = dict(style=np.array(style), box_lo=s.box.lo, box_prd=s.box.prd, box_tilt=s.box.tilt, x=s.x,
             type=s.type.astype(np.int8), tag=s.tag, mass=s.mass)
    idx = sample_index(s)
    for k, v in out.items():
        if idx is not None and k in PER_ATOM:
            d["out_" + k] = v[idx]
            d["sum_" + k] = np.asarray(v, dtype=np.float64).sum(axis=0)
        elif idx is not None and k == "f_raw":
            continue                               # (the ghost shell of a 32 000-atom cell: not stored)
        else:
            d["out_" + k] = v
    if idx is not None:
        d["sample"] = idx
    return d


def load(name):
    z = np.load(os.path.join(FIXDIR, name + ".npz"))
    s = S.System(S.Box(z["box_lo"], z["box_prd"], z["box_tilt"]), np.ascontiguousarray(z["x"]),
                 z["type"].astype(np.int32), z["tag"].astype(np.int32), z["mass"])
    out = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
    sums = {k[4:]: z[k] for k in z.files if k.startswith("sum_")}
    return str(z["style"]), s, out, (z["sample"] if "sample" in z.files else None), sums


def names():
    return sorted(list(CASES) + [f"R-bulk-0-step{k}" for k in BULK_STEPS])


def bulk_states(orc, P):
    """positions of the in.rebomos-bulk run at steps 0, 10, 20 (fix nve around the oracle)"""
    s = S.rebomos_bulk_cell()
    eng = mdref.RebomosCPU(orc, P, s)
    m = s.mass[s.type][:, None]
    x, v = s.x.copy(), np.zeros_like(s.x)
    f = eng.compute(x)["f_owned"]
    states = {0: x.copy()}
    for step in range(1, max(BULK_STEPS) + 1):
        v += 0.5 * 0.001 * S.FTM2V * f / m
        x += 0.001 * v
        f = eng.compute(x)["f_owned"]
        v += 0.5 * 0.001 * S.FTM2V * f / m
        if step in BULK_STEPS:
            states[step] = x.copy()
    return s, {k: dataclasses.replace(s, x=xs) for k, xs in states.items()}
