#!/usr/bin/env python3
"""PCIe-inclusive rate of the HOST-MODE boundary (what a LAMMPS Pair::compute() sees): per step the host
uploads x of owned+ghost atoms (24 B/atom), the device computes, forces of owned atoms come back (24 B/atom).
Atoms are handed over in lattice (creation) order, as a host would; the library sorts its own copy.
Not `bench.py`'s `value` (that one is device-resident); the number is quoted in DESIGN.md section 7.
usage: python profiles/host_mode_rate.py [nx ny nz] [steps]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry

entry.load_package()
from lammps_plugins_amd.host import capi, system as S

rep = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (24, 24, 24)
steps = int(sys.argv[4]) if len(sys.argv) >= 5 else 20
pot = os.path.join(ROOT, "tests", "golden", "potentials", "MoS.REBO.set5b")
s = S.replicate(S.rebomos_bulk_cell(), rep)
p = capi.read_rebomos_file(pot)
skin = 2.0
cutghost = 3.0 * p.rcmax[0][0] + skin
xw = S.wrap(s.box, s.x)
owner, shift = S.make_ghosts(s.box, xw, cutghost)
xa = np.ascontiguousarray(np.concatenate([xw, xw[owner] + shift @ s.box.h.T]))
type_all = np.concatenate([s.type, s.type[owner]]).astype(np.int32)
tag_all = np.concatenate([s.tag, s.tag[owner]]).astype(np.int32)
n = s.n
ctx = capi.Context(0)
ctx.rebomos_set_params(p)
t0 = time.perf_counter()
ctx.set_atoms_host(n, xa, type_all, tag_all, 2, map_=[0, 0, 1])
ctx.set_skin(skin)
f = np.zeros((n, 3))
eng, vir = capi.C.c_double(0.0), np.zeros(6)


def compute():
    ctx._ck(ctx.L.mdp_rebomos_compute_host(ctx.h, 0, 0, capi._dp(f), capi.C.byref(eng), capi._dp(vir), None, None))


compute()                       # builds the style's lists
t_first = time.perf_counter() - t0
for _ in range(3):
    ctx.set_positions_host(xa)
    compute()
t0 = time.perf_counter()
for _ in range(steps):
    ctx.set_positions_host(xa)
    compute()
dt = (time.perf_counter() - t0) / steps
info = ctx.rebomos_list_info()
print(json.dumps({"atoms": n, "ghosts": int(len(owner)), "ms_per_step_host_mode": round(dt * 1e3, 3),
                  "Matom_steps_per_s_pcie_inclusive": round(n / dt / 1e6, 2), "first_call_s": round(t_first, 2),
                  "tiled": info["tiled"], "union_max": info["union_max"],
                  "bytes_up_per_step": int(xa.nbytes), "bytes_down_per_step": int(f.nbytes)}))
ctx.close()
