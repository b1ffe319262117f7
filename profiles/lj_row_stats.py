#!/usr/bin/env python3
"""Row statistics of the Lennard-Jones tile lists the kernel walks (REBO-MoS bulk, replicate N N N, default 24):
entries per two-atom row as built and as pruned, 16-lane steps per row, and -- counted on the host for a sample of
tiles from the same positions -- how many (entry, atom) evaluations lie inside a window.  usage: lj_row_stats.py [N] [T]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from lammps_plugins_amd.host import capi, resident, system as S  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
temp = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
POT = os.path.join(ROOT, "tests", "golden", "potentials", "MoS.REBO.set5b")
s = S.replicate(S.rebomos_bulk_cell(), (n, n, n))
v0 = S.gaussian_velocities(s, temp, seed=1082337) if temp > 0 else None
ctx = capi.Context(0)
pot = capi.read_rebomos_file(POT)
ctx.rebomos_set_params(pot)
dom = resident.DeviceDomain(ctx, capi.STYLE_REBOMOS, s, 3.0 * pot.rcmax[0][0] + 2.0, 2.0, [0, 0, 1], v0=v0)
dom.compute(1, 1)
for k in range(12):
    dom.step(0, 0, rebuild="auto", defer_final=True)
dom.flush()
info = ctx.rebomos_list_info()
ncl = info["clusters"]
ln = ctx.md_download_int("lj_len", ncl).astype(np.int64)
sp = ctx.md_download_int("lj_split", ncl).astype(np.int64)
typ = ctx.md_download_int("type", dom.nlocal)
out = dict(atoms=s.n, clusters=int(ncl), tiles=int(info["tiles"]), union_max=int(info["union_max"]),
           entries_per_row_as_built=info["row_entries"] / ncl,
           entries_per_row_pruned=float(ln.mean()), first_segment_pruned=float(sp.mean()),
           trips_per_row_pruned=float(ln.mean() / 16.0),
           evaluations_per_atom_pruned=float(ln.sum() * 2 / dom.nlocal),
           in_window_per_atom_reference=127.3,
           prune=ctx.md_prune_stats())
el = np.array([0, 0, 1])[typ[:2 * ncl:2]] if dom.nlocal >= 2 * ncl else None
if el is not None:
    out["entries_per_row_pruned_Mo_cluster"] = float(ln[el == 0].mean())
    out["entries_per_row_pruned_S_cluster"] = float(ln[el == 1].mean())
print(json.dumps(out, indent=1))
ctx.close()
