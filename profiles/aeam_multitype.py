#!/usr/bin/env python3
"""AEAM with more than two atom types, resident mode: ms per step of the same 1 M-atom alloy (config #3: fcc Al,
0.75 % Si, 863 K) labelled three ways
  two types            the bundled AlSi.aeam, specialised two-type tile kernels + persistent density kernel
  five types, tiles    a five-element file made of the same functions (3 metals, 2 angular); tile lists with two
                       segments (type 0 | the other four), per-entry types and parameters from LDS
  five types, CSR      the same with MDP_AEAM_TILE=0: per-atom CSR lists and the list-streaming kernels
usage: python profiles/aeam_multitype.py [ncell] [steps]"""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry

entry.load_package()
import aeam_five
from lammps_plugins_amd.host import capi, resident, system as S

ncell = int(sys.argv[1]) if len(sys.argv) > 1 else 63
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
pot = os.path.join(ROOT, "tests", "golden", "potentials", "AlSi.aeam")
five = os.path.join(tempfile.mkdtemp(), "five.aeam")
aeam_five.write_five_element_file(five, pot)

s2 = S.fcc_cell(4.045, ncell, frac_type2=0.0075, seed=7683797)
rng = np.random.default_rng(3)
t5 = np.where(s2.type == 1, rng.integers(1, 4, s2.n), rng.integers(4, 6, s2.n)).astype(np.int32)
v0 = S.gaussian_velocities(s2, 863.0, seed=4928459)
out = {"atoms": int(s2.n), "steps": steps}
for tag, path, types, tile in (("two types", pot, s2.type, "1"), ("five types, tiles", five, t5, "1"),
                               ("five types, CSR", five, t5, "0")):
    os.environ["MDP_AEAM_TILE"] = tile
    af = capi.AeamFile(path)
    tabs = af.build()
    s = S.System(s2.box, s2.x.copy(), types.copy(), s2.tag.copy(), np.array([0.0] + list(af.mass)))
    ctx = capi.Context(0)
    ctx.aeam_set_tables(tabs)
    d = resident.DeviceDomain(ctx, capi.STYLE_AEAM, s, float(af.cut_table(tabs).max()) + 1.0, 1.0, None, v0=v0)
    d.compute(0, 0)
    for _ in range(20):
        d.step(0, 0, rebuild="auto")
    ctx.sync()
    b0 = d.builds
    t0 = time.perf_counter()
    for _ in range(steps):
        d.step(0, 0, rebuild="auto")
    ctx.sync()
    dt = time.perf_counter() - t0
    d.compute(1, 0)
    th = d.thermo()
    out[tag] = {"ms_per_step": round(dt / steps * 1e3, 4), "Matom_steps_per_s": round(s.n * steps / dt / 1e6, 1),
                "reneighborings": d.builds - b0, "pe_per_atom": th["pe"] / s.n, "temp": th["temp"]}
    ctx.close()
print(json.dumps(out))
