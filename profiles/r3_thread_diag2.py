#!/usr/bin/env python3
"""Diagnosis of the 8-brick ownership failure (tests/test_gpu_domain.py::test_eight_bricks_hot_run...): the same run
with and without the process-wide library lock, and with one brick; prints how many atoms are owned 0 / 2 times and
after which reneighboring.  usage: python3 profiles/r3_thread_diag2.py <serialize 0|1> [world] [nrep]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MDP_THREAD_SERIALIZE"] = sys.argv[1] if len(sys.argv) > 1 else "0"
os.environ["MDP_DIAG"] = "1"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nrep = int(sys.argv[3]) if len(sys.argv) > 3 else 12
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
import numpy as np  # noqa: E402

from lammps_plugins_amd.host import capi, resident, system as S  # noqa: E402

POT = os.path.join(ROOT, "tests", "golden", "potentials", "MoS.REBO.set5b")
s = S.replicate(S.rebomos_bulk_cell(), (nrep, nrep, nrep))
v0 = S.gaussian_velocities(s, 300.0, seed=23) + np.array([50.0, -35.0, 20.0])
log = {}


def rank_fn(r, make_tr):
    ctx = capi.Context(0)
    p = capi.read_rebomos_file(POT)
    ctx.rebomos_set_params(p)
    tr = make_tr(ctx) if world > 1 else None
    d = resident.DeviceDomain(ctx, capi.STYLE_REBOMOS, s, 3.0 * p.rcmax[0][0] + 2.0, 2.0, [0, 0, 1], v0=v0, transport=tr)
    hist = [(0, d.nlocal, d.tags_local.copy())]
    d.compute(0, 0)
    for step in range(1, 31):
        rb = step % 10 == 0
        d.step(0, 0, rebuild=rb)
        if rb:
            hist.append((step, d.nlocal, d.tags_local.copy()))
    ctx.close()
    return hist


res = resident.run_ranks(world, rank_fn) if world > 1 else [rank_fn(0, None)]
for k in range(len(res[0])):
    seen = np.zeros(s.n, dtype=int)
    for h in res:
        seen[h[k][2] - 1] += 1
    print(f"serialize={os.environ['MDP_THREAD_SERIALIZE']} world={world} after step {res[0][k][0]}: nlocal {[h[k][1] for h in res]} "
          f"sum {sum(h[k][1] for h in res)} of {s.n}; owned 0x: {(seen == 0).sum()}, 2x: {(seen == 2).sum()}, >2x: {(seen > 2).sum()}",
          flush=True)
