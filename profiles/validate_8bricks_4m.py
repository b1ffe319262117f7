#!/usr/bin/env python3
"""Production-size check of the 8-GPU decomposition on ONE GPU: the 3,981,312-atom REBO-MoS bulk on 2x2x2 bricks
(eight rank threads, resident.ThreadTransport) against the same run on one brick.  300 K plus a uniform drift, so
that thousands of atoms change owner at the forced reneighborings.  Prints the largest position / velocity
difference per atom tag and the thermo rows of both runs.
usage: python profiles/validate_8bricks_4m.py [nrep] [steps] [rebuild_every]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry

entry.load_package()
from lammps_plugins_amd.host import capi, resident, system as S

nrep = int(sys.argv[1]) if len(sys.argv) > 1 else 24
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
every = int(sys.argv[3]) if len(sys.argv) > 3 else 10
style = sys.argv[4] if len(sys.argv) > 4 else "rebomos"      # or "aeam": fcc nrep^3 cells, 0.75 % Si, 863 K
pot = os.path.join(ROOT, "tests", "golden", "potentials", "MoS.REBO.set5b")
pot_aeam = os.path.join(ROOT, "tests", "golden", "potentials", "AlSi.aeam")
if style == "rebomos":
    s = S.replicate(S.rebomos_bulk_cell(), (nrep, nrep, nrep))
    v0 = S.gaussian_velocities(s, 300.0, seed=11) + np.array([60.0, -45.0, 30.0])
else:
    s = S.fcc_cell(4.045, nrep, frac_type2=0.0075, seed=7683797)
    s.mass[1:3] = capi.AeamFile(pot_aeam).mass[:2]
    v0 = S.gaussian_velocities(s, 863.0, seed=11) + np.array([60.0, -45.0, 30.0])


def run(world):
    def rank_fn(r, make_tr):
        ctx = capi.Context(0)
        if style == "rebomos":
            p = capi.read_rebomos_file(pot)
            ctx.rebomos_set_params(p)
            args = (capi.STYLE_REBOMOS, s, 3.0 * p.rcmax[0][0] + 2.0, 2.0, [0, 0, 1])
        else:
            af = capi.AeamFile(pot_aeam)
            tabs = af.build()
            ctx.aeam_set_tables(tabs)
            ctx._keep = (af, tabs)
            args = (capi.STYLE_AEAM, s, float(af.cut_table(tabs).max()) + 1.0, 1.0, None)
        tr = make_tr(ctx) if world > 1 else None
        d = resident.DeviceDomain(ctx, *args, v0=v0, transport=tr)
        d.compute(1, 1)
        th0 = d.thermo()
        left = 0
        for k in range(1, steps + 1):
            rb = k % every == 0
            d.step(1 if k == steps else 0, 1 if k == steps else 0, rebuild=rb)
            if rb:
                left += ctx.dd_info()["left_last"]
        th = d.thermo()
        got = ctx.md_download(d.nlocal, want=("x", "v"))
        tags = d.tags_local
        ctx.close()
        return tags, got["x"], got["v"], th0, th, left, (d.nlocal, d.nself, d.nrecv)

    t0 = time.perf_counter()
    res = [rank_fn(0, None)] if world == 1 else resident.run_ranks(world, rank_fn)
    wall = time.perf_counter() - t0
    x, v = np.zeros((s.n, 3)), np.zeros((s.n, 3))
    for tags, xx, vv, *_ in res:
        x[tags - 1], v[tags - 1] = xx, vv
    return dict(x=x, v=v, th0=res[0][3], th=res[0][4], left=sum(r[5] for r in res), counts=[r[6] for r in res], wall=wall)


a, b = run(1), run(8)
dx = b["x"] - a["x"]
dx -= np.round(s.box.x2lamda(dx + s.box.lo)) @ s.box.h.T
print(json.dumps({
    "style": style, "atoms": s.n, "steps": steps, "reneighbor_every": every, "atoms_that_changed_owner": int(b["left"]),
    "bricks_nlocal_nself_nremote": b["counts"],
    "max_dx_A": float(np.abs(dx).max()), "max_dv_A_per_ps": float(np.abs(b["v"] - a["v"]).max()),
    "pe_start_1_vs_8": [a["th0"]["pe"], b["th0"]["pe"]], "pe_end_1_vs_8": [a["th"]["pe"], b["th"]["pe"]],
    "ke_end_1_vs_8": [a["th"]["ke"], b["th"]["ke"]], "press_end_1_vs_8": [a["th"]["press"], b["th"]["press"]],
    "wall_s_1_vs_8_sharing_one_gpu": [round(a["wall"], 1), round(b["wall"], 1)]}))
