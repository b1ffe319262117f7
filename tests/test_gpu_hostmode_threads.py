"""GPU: the drop-in (host-mode) path called from two threads of one process, one context each, at a size where the
library's host-side worker threads take the staging copies and the result adds (> 4 MB of positions, > 1 M force
components).  Only one caller at a time gets the workers, the other does its pieces itself: both must return the
forces of the 288-atom cell replicated -- every replica alike -- and equal each other bit for bit."""
import threading

import numpy as np
import pytest

from conftest import POT_REBOMOS
from lammps_plugins_amd.host import capi, system as S

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(600)
def test_two_contexts_compute_in_host_mode_concurrently():
    p = capi.read_rebomos_file(POT_REBOMOS)
    cell = S.rebomos_bulk_cell()
    s = S.replicate(cell, (12, 12, 12))                   # 497 664 atoms: 1.49 M force components
    cutghost = 3.0 * p.rcmax[0][0] + 2.0
    xw = S.wrap(s.box, s.x)
    owner, shift = S.make_ghosts(s.box, xw, cutghost)
    xa = np.ascontiguousarray(np.concatenate([xw, xw[owner] + shift @ s.box.h.T]))      # (main thread: BLAS is fine here)
    type_all = np.concatenate([s.type, s.type[owner]]).astype(np.int32)
    tag_all = np.concatenate([s.tag, s.tag[owner]]).astype(np.int32)
    n = s.n
    assert xa.nbytes > (4 << 20) and 3 * n > (1 << 20)
    results, errors = [None, None], []

    def rank(k):
        try:
            ctx = capi.Context(0)
            ctx.rebomos_set_params(p)
            ctx.set_atoms_host(n, xa, type_all, tag_all, 2, map_=[0, 0, 1])
            ctx.set_skin(2.0)
            out = []
            for rep in range(3):                          # the calls of the two threads interleave
                f = np.full((n, 3), 1.0 + k)              # results are ADDED to what the host holds
                eng, vir = capi.C.c_double(0.0), np.zeros(6)
                ctx.set_positions_host(xa)
                ctx._ck(ctx.L.mdp_rebomos_compute_host(ctx.h, 1, 0, capi._dp(f), capi.C.byref(eng), capi._dp(vir),
                                                       None, None))
                out.append((f - (1.0 + k), eng.value))
            ctx.close()
            results[k] = out
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=rank, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    f0, e0 = results[0][0]
    for k in range(2):
        for f, e in results[k]:
            assert np.abs(f - f0).max() < 1e-12           # (the offset 1 + k is added and subtracted in FP64)
            assert e == pytest.approx(e0, rel=1e-13)
    # every replica of the cell carries the forces of the cell
    fc = f0.reshape(12 ** 3, cell.n, 3)
    assert np.abs(fc - fc[0]).max() < 1e-9
    assert e0 / n == pytest.approx(-7.158372, abs=5e-7)
