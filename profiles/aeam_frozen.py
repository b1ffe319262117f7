"""Kernel times of the AEAM path on a FROZEN configuration: configuration #3 (1 000 188 atoms, 0.75 % Si) is run for a few
hundred NVE steps from 863 K, then the positions stay where they are and compute() is repeated -- so that timing-only
variants of a kernel (debug switches that leave forces wrong) can be compared on the same rows, unions and disorder.
usage: python3 profiles/aeam_frozen.py [md_steps=150] [repeats=40] [ncell=63]   -> one JSON line"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from lammps_plugins_amd.host import capi, resident, system as S  # noqa: E402

md_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ncell = int(sys.argv[3]) if len(sys.argv) > 3 else 63
POT = os.path.join(ROOT, "tests", "golden", "potentials", "AlSi.aeam")
s = S.fcc_cell(4.045, (ncell,) * 3, frac_type2=0.0075, seed=7683797)
af = capi.AeamFile(POT)
tabs = af.build()
s.mass[1:3] = af.mass[:2]
v0 = S.gaussian_velocities(s, 863.0, seed=1082337)
ctx = capi.Context(0)
ctx.aeam_set_tables(tabs)
cutghost = float(af.cut_table(tabs).max()) + 1.0
env_dbg = os.environ.pop("MDP_PF_DBG", None)          # the MD part runs with the real kernel
d = resident.DeviceDomain(ctx, capi.STYLE_AEAM, s, cutghost, 1.0, None, v0=v0)
d.compute(0, 0)
for step in range(md_steps):
    d.step(0, 0, rebuild="auto", defer_final=True)
d.flush()
if env_dbg is not None:
    os.environ["MDP_PF_DBG"] = env_dbg
for _ in range(3):
    ctx.md_compute(0, 0)
ctx.sync()
ctx.set_timing(True)
acc = np.zeros(8)
for _ in range(repeats):
    ctx.md_compute(0, 0)
    acc += np.array(ctx.get_timing())
ctx.set_timing(False)
acc /= repeats
th = d.thermo()
names = ["density", "density_ang", "embed", "force", "force_ang", "density_int", "force_int", "-"]
print(json.dumps(dict(atoms=s.n, md_steps=md_steps, temp=round(th["temp"], 1), prune=ctx.md_prune_stats(),
                      ms={n: round(float(v), 4) for n, v in zip(names, acc) if v})))
ctx.close()
