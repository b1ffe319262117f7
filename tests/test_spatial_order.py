"""CPU tests of the storage-order helpers (host/resident.py): the Lennard-Jones tile lists rely on ANY run of
consecutive atoms being a compact blob."""
import numpy as np

from lammps_plugins_amd.host import resident, system as S, order


def _cells(n):
    g = np.arange(n)
    return np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3).astype(float) + 0.5


def test_hilbert_curve_has_no_jumps_on_a_full_cube():
    x = _cells(16)
    o = order.hilbert_order(x, np.zeros(3), 1.0)
    assert sorted(o) == list(range(len(x)))
    step = np.abs(np.diff(x[o], axis=0)).sum(axis=1)
    assert step.max() == 1.0                         # consecutive cells share a face
    zo = order.morton_order(x, np.zeros(3), 1.0)
    assert np.abs(np.diff(x[zo], axis=0)).sum(axis=1).max() > 10   # the Z-order curve jumps


def test_runs_of_consecutive_atoms_are_compact_in_a_triclinic_box():
    """The in.rebomos-bulk cell is triclinic (xy tilt of half an edge).  The curve must (a) be fitted to the atoms'
    extent -- a partly occupied 2^k cube brings jumps back -- and (b) run in lamda coordinates -- in Cartesian
    coordinates the atoms fill a parallelepiped inside their bounding box and the curve crosses its empty corners.
    Runs of 32 consecutive atoms (one Lennard-Jones tile) then stay compact."""
    s = S.replicate(S.rebomos_bulk_cell(), (5, 5, 4))
    x = S.wrap(s.box, s.x)

    def extents(order):
        xs = x[order]
        # minimum-image extent of each run, measured in lamda space and scaled back to lengths
        lam = s.box.x2lamda(xs) * np.linalg.norm(s.box.h, axis=0)
        return np.array([np.ptp(lam[k:k + 32], axis=0).max() for k in range(0, len(xs) - 32, 32)])

    e_box = extents(order.spatial_order(x, s.box.lo, 3.0, box=s.box))
    e_cart = extents(order.hilbert_order(x, s.box.lo, 3.0))
    e_z = extents(order.morton_order(x, s.box.lo, 3.0))
    assert e_box.max() < 30.0
    assert e_box.max() < e_cart.max() < e_z.max()
    assert e_box.mean() <= e_z.mean()


def test_spatial_order_groups_types_inside_stretches(monkeypatch):
    s = S.replicate(S.rebomos_bulk_cell(), (3, 3, 2))
    x = S.wrap(s.box, s.x)
    monkeypatch.setenv("MDP_ORDER_GROUP", "1")
    o = order.spatial_order(x, s.box.lo, 3.0, group=s.type, chunk=96)
    assert sorted(o) == list(range(s.n))
    t = s.type[o]
    for k in range(0, s.n, 96):
        assert np.all(np.diff(t[k:k + 96]) >= 0)     # sorted by type inside every stretch of the curve
    monkeypatch.setenv("MDP_ORDER_GROUP", "0")
    o0 = order.spatial_order(x, s.box.lo, 3.0, group=s.type)
    assert np.array_equal(o0, order.hilbert_order(x, s.box.lo, 3.0))


def test_empty_input():
    assert len(order.hilbert_order(np.zeros((0, 3)), np.zeros(3), 3.0)) == 0


def test_box_transforms_are_safe_in_rank_threads():
    """resident.run_ranks runs the ranks of a rehearsal as threads, and every rank wraps the global system and picks
    its brick with numpy.  `a @ b.T` goes to the BLAS library and returned wrong rows now and then when eight threads
    did it at once (round-2's spurious 'more than 64 neighbours', round-3's atoms owned twice); Box.x2lamda /
    lamda2x are written out element by element.  Eight threads must reproduce the single-thread result exactly."""
    import threading
    s = S.replicate(S.rebomos_bulk_cell(), (5, 5, 4))
    ref_w = S.wrap(s.box, s.x)
    ref_l = s.box.x2lamda(ref_w)
    bad = [0] * 8

    def work(r):
        for _ in range(25):
            w = S.wrap(s.box, s.x)
            if not (np.array_equal(w, ref_w) and np.array_equal(s.box.x2lamda(w), ref_l)):
                bad[r] += 1

    th = [threading.Thread(target=work, args=(r,)) for r in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert bad == [0] * 8
    assert np.abs(s.box.hinv @ s.box.h - np.eye(3)).max() < 1e-15
