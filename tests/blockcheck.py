"""Parity at scale (tests only): forces of a multi-million-atom DISORDERED run meet the CPU oracle.

The oracle cannot run four or sixteen million atoms, and the full-size checks through extensive properties use
states in which every replica of the cell moves alike.  Here the positions of a hot run are downloaded, blocks of
~500 atoms are cut out together with everything within `shell` of them (minimum image), each block + shell is run
through the oracle as an isolated cluster in a large box, and the forces of the block's INTERIOR atoms -- whose whole
interaction range lies inside the cluster -- are compared with what the device computed for the same atoms in the
full system.  Blocks are placed where the device machinery has its seams: box corners (periodic images), brick
faces and corners of the 2 x 2 x 2 decomposition, the tile with the largest neighbour union, the last (partly
filled) tile, and random places."""
from __future__ import annotations

import numpy as np

from lammps_plugins_amd.host import system as S


def cut_block(box: S.Box, x: np.ndarray, point: np.ndarray, n_interior: int, shell: float):
    """indices of the n_interior atoms nearest to `point` and of all atoms within `shell` of that ball, plus the
    minimum-image displacements of the latter from `point`"""
    lam = S.mul_upper(x - point, box.hinv)
    lam -= np.round(lam)
    d = S.mul_upper(lam, box.h)
    r2 = np.einsum("ij,ij->i", d, d)
    inner = np.argpartition(r2, n_interior)[:n_interior]
    r_int = float(np.sqrt(r2[inner].max()))
    half = 0.5 * min(box.prd[0], box.prd[1], box.prd[2])
    assert r_int + shell < 0.45 * half, "block too large for the minimum-image construction"
    env = np.nonzero(r2 <= (r_int + shell) ** 2)[0]
    return inner, env, d[env], r_int


def cluster_system(d_env: np.ndarray, type_env: np.ndarray, tag_env: np.ndarray, mass: np.ndarray, margin: float):
    """the block + shell as an isolated cluster: a cubic box so large that no periodic image is in range"""
    ext = float(np.abs(d_env).max())
    L = 2.0 * (ext + margin)
    box = S.Box(np.zeros(3), np.array([L, L, L]), np.zeros(3))
    return S.System(box, np.ascontiguousarray(d_env + 0.5 * L), type_env.astype(np.int32), tag_env.astype(np.int32), mass)


def seeds(box: S.Box, x_dev: np.ndarray, extra_points=(), n_random=2, seed=5):
    """where to cut: box corners, seams of a 2x2x2 brick decomposition, the last atom in device order, random places"""
    rng = np.random.default_rng(seed)
    lam = [np.array([0.0, 0.0, 0.0]), np.array([1.0, 1.0, 1.0]) - 1e-9, np.array([0.5, 0.5, 0.5]),
           np.array([0.5, 0.25, 0.75]), np.array([0.25, 0.5, 0.0])]
    pts = [box.lamda2x(l) for l in lam]
    pts.append(x_dev[-1].copy())
    pts += [np.asarray(p, dtype=float) for p in extra_points]
    pts += [box.lamda2x(rng.random(3)) for _ in range(n_random)]
    return pts


def check_blocks(box, x_dev, f_dev, type_dev, tag_dev, mass, points, engine_factory, n_interior, shell, margin, tol):
    """engine_factory(System) -> object with .compute(x) returning dict(f_owned=...).  Returns the worst deviation."""
    worst, rows = 0.0, []
    for p in points:
        inner, env, d_env, r_int = cut_block(box, x_dev, p, n_interior, shell)
        cs = cluster_system(d_env, type_dev[env], tag_dev[env], mass, margin)
        eng = engine_factory(cs)
        assert eng.nghost == 0                       # really isolated
        o = eng.compute(cs.x, eflag=0, vflag=0)
        pos = {int(g): k for k, g in enumerate(env)}
        sel = np.array([pos[int(g)] for g in inner])
        dev = float(np.abs(o["f_owned"][sel] - f_dev[inner]).max())
        rows.append((len(env), r_int, dev))
        worst = max(worst, dev)
    assert worst < tol, f"interior forces differ from the oracle by {worst:.3e} (blocks: {rows})"
    return worst, rows
