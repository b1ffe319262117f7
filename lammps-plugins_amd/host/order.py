"""Storage order of the atoms (host side): Z-order and Hilbert curves, element-grouped stretches.  The device keys
its atoms itself at every reneighboring (csrc/domain.hip, csrc/mdp_api.hip); these functions serve the host-planned
reference decomposition (host/decomp.py, tests/hostplan.py) and the tests of the ordering itself."""
from __future__ import annotations

import numpy as np


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


def hilbert_order(x: np.ndarray, lo: np.ndarray, cell: float) -> np.ndarray:
    """argsort of atoms along a 3-D Hilbert curve on a `cell`-sized grid.  Unlike the Z-order curve a
    Hilbert curve has no jumps: ANY run of consecutive atoms is a compact blob, which is what bounds the
    neighbour union of the Lennard-Jones tiles (csrc/rebomos.hip) and with it their LDS footprint.
    (Skilling's axes-to-transpose algorithm, vectorised.)"""
    if len(x) == 0:
        return np.zeros(0, dtype=np.int64)
    # exactly 2^bits cells per dimension over the atoms' extent (cells need not be cubic): the curve is
    # continuous only on its full cube, a partly occupied cube would bring the jumps back
    xmin = x.min(axis=0)
    ext = np.maximum(x.max(axis=0) - xmin, 1e-9)
    bits = max(1, int(np.ceil(np.log2(max(ext.max() / cell, 1.0)))))
    ncell = 1 << bits
    g = np.minimum(np.floor((x - xmin) / ext * ncell).astype(np.int64), ncell - 1)
    X = [g[:, 0].astype(np.uint64), g[:, 1].astype(np.uint64), g[:, 2].astype(np.uint64)]
    zero = np.uint64(0)
    q = 1 << (bits - 1)
    while q > 1:
        Q, P = np.uint64(q), np.uint64(q - 1)
        for i in range(3):
            hit = (X[i] & Q) != zero
            X[0] = np.where(hit, X[0] ^ P, X[0])
            t = np.where(hit, zero, (X[0] ^ X[i]) & P)
            X[0] = X[0] ^ t
            X[i] = X[i] ^ t
        q >>= 1
    X[1] ^= X[0]
    X[2] ^= X[1]
    t = np.zeros_like(X[0])
    q = 1 << (bits - 1)
    while q > 1:
        t = np.where((X[2] & np.uint64(q)) != zero, t ^ np.uint64(q - 1), t)
        q >>= 1
    X = [v ^ t for v in X]
    key = np.zeros_like(X[0])
    for b in range(bits - 1, -1, -1):
        for i in range(3):
            key = (key << np.uint64(1)) | ((X[i] >> np.uint64(b)) & np.uint64(1))
    return np.argsort(key, kind="stable")


def spatial_order(x: np.ndarray, lo: np.ndarray, cell: float, group=None, chunk: int = 96, box=None) -> np.ndarray:
    """The order atoms are stored in on the device: along a Hilbert curve (MDP_ORDER=morton: Z-order), and,
    when `group` (the atom types) is given, each stretch of `chunk` consecutive atoms additionally sorted by
    type.  The second step makes the 2-atom clusters and 32-atom tiles of the Lennard-Jones lists
    element-pure: pair cutoffs differ per element pair (Mo-Mo 10.5 A, S-S 7.8 A), a mixed cluster evaluates
    both atoms against the larger neighbourhood, and the four clusters sharing a wavefront all run as long as
    the longest list among them.  chunk = 3 tiles keeps every tile inside one compact stretch of the curve."""
    import os
    if box is not None:
        # Order in lamda (fractional) coordinates, rescaled to the edge lengths: in a TRICLINIC box (the
        # in.rebomos-bulk cell has an xy tilt of half an edge) the atoms fill a parallelepiped inside their
        # Cartesian bounding box and a curve over that box crosses its empty corners -- consecutive atoms jump.
        x = box.x2lamda(x) * np.linalg.norm(box.h, axis=0)
        lo = np.zeros(3)
    order = morton_order(x, lo, cell) if os.environ.get("MDP_ORDER", "hilbert") == "morton" else hilbert_order(x, lo, cell)
    if group is not None and chunk > 0 and len(order) and os.environ.get("MDP_ORDER_GROUP", "0") != "0":
        g = np.asarray(group)[order].astype(np.int64)
        key = (np.arange(len(order), dtype=np.int64) // chunk) * (int(g.max()) + 1) + g
        order = order[np.argsort(key, kind="stable")]
    return order
