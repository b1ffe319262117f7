#!/usr/bin/env python3
"""NVE energy conservation over a longer run on the device domain: device lists rebuilt by the style's own trigger,
reneighboring (remap, ghosts, lists) by the deferred on-device `check yes` flag.
usage: python profiles/nve_drift.py [nrep] [steps] [T] [dt_ps]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from lammps_plugins_amd.host import capi, resident, system as S

nrep = int(sys.argv[1]) if len(sys.argv) > 1 else 12
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
T = float(sys.argv[3]) if len(sys.argv) > 3 else 300.0
dt = float(sys.argv[4]) if len(sys.argv) > 4 else 0.001
s = S.replicate(S.rebomos_bulk_cell(), (nrep, nrep, nrep))
ctx = capi.Context(0)
p = capi.read_rebomos_file(os.path.join(ROOT, "tests", "golden", "potentials", "MoS.REBO.set5b"))
ctx.rebomos_set_params(p)
skin = 2.0
cutghost = 3.0 * p.rcmax[0][0] + skin
v0 = S.gaussian_velocities(s, T, seed=12345)
d = resident.DeviceDomain(ctx, capi.STYLE_REBOMOS, s, cutghost, skin, [0, 0, 1], v0=v0, dt=dt)
d.compute(1, 0)
t = d.thermo()
e0 = t["pe"] + t["ke"]
rows = [(0, t["pe"], t["ke"], 0.0)]
b0 = d.builds
for step in range(1, steps + 1):
    ev = 1 if step % 100 == 0 else 0
    d.step(ev, 0, rebuild="auto")
    if ev:
        t = d.thermo()
        rows.append((step, t["pe"], t["ke"], (t["pe"] + t["ke"] - e0) / s.n))
info = ctx.rebomos_list_info()
print(json.dumps({"atoms": s.n, "steps": steps, "T0": T, "dt_ps": dt, "style_list_builds": info["builds"],
                  "reneighborings": d.builds - b0, "dangerous": d.dangerous,
                  "drift_eV_per_atom": [round(r[3], 9) for r in rows], "temp_end": round(S.temperature(rows[-1][2], s.n), 2)}))
ctx.close()
