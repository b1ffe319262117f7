#!/usr/bin/env python3
"""CPU cost of driving one resident-mode step (Python harness -> ctypes -> kernel launches), measured on systems so
small that the GPU work per step is negligible: the step rate is then bound by the host side alone.
usage: python profiles/step_cpu_cost.py [steps]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry

entry.load_package()
from lammps_plugins_amd.host import capi, resident, system as S

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
out = {}
pot = os.path.join(ROOT, "tests", "golden", "potentials", "AlSi.aeam")
af = capi.AeamFile(pot)
tabs = af.build()
s = S.fcc_cell(4.045, 10, frac_type2=0.0075, seed=7683797)
s.mass[1:3] = af.mass
ctx = capi.Context(0)
ctx.aeam_set_tables(tabs)
d = resident.DeviceDomain(ctx, capi.STYLE_AEAM, s, float(af.cut_table(tabs).max()) + 1.0, 1.0, None,
                          v0=S.gaussian_velocities(s, 300.0, seed=1))
d.compute(0, 0)
for tag, kw in (("aeam, check every step", dict(rebuild="auto")), ("aeam, no check", dict(rebuild=False))):
    for _ in range(50):
        d.step(0, 0, **kw)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        d.step(0, 0, **kw)
    ctx.sync()
    out[tag] = {"atoms": int(s.n), "us_per_step": round((time.perf_counter() - t0) / steps * 1e6, 1)}
ctx.close()

p = capi.read_rebomos_file(os.path.join(ROOT, "tests", "golden", "potentials", "MoS.REBO.set5b"))
s = S.replicate(S.rebomos_bulk_cell(), (2, 2, 2))
ctx = capi.Context(0)
ctx.rebomos_set_params(p)
d = resident.DeviceDomain(ctx, capi.STYLE_REBOMOS, s, 3.0 * p.rcmax[0][0] + 2.0, 2.0, [0, 0, 1])
d.compute(0, 0)
for tag, kw in (("rebomos, check every step", dict(rebuild="auto")), ("rebomos, no check", dict(rebuild=False))):
    for _ in range(50):
        d.step(0, 0, **kw)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        d.step(0, 0, **kw)
    ctx.sync()
    out[tag] = {"atoms": int(s.n), "us_per_step": round((time.perf_counter() - t0) / steps * 1e6, 1)}
ctx.close()
print(json.dumps(out))
