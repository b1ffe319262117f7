#!/usr/bin/env python3
"""Device-resident trajectories against the same trajectories driven on the host with ORACLE forces (the helpers of
tests/test_gpu_trajectory.py, longer runs, numbers written out): worst position difference and worst total-energy
difference per atom over the samples, and the energy series of both sides -- the evidence that the energy drift of a
device run is the potential's own.  usage: python3 profiles/trajectory_pin.py [steps, default 1000] > r04_trajectory_pin.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry

entry.load_package()
from lammps_plugins_amd.host import capi, system as S
import mdref
import oracle_bindings as ob
import test_gpu_trajectory as T
from conftest import POT_AEAM, POT_REBOMOS

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
every = 50
orc = ob.load()
out = {}

os.environ["MDP_INNER_SKIN"] = "0.5"
P = orc.rebomos_params(POT_REBOMOS)
s = S.replicate(S.rebomos_bulk_cell(), (2, 2, 2))
v0 = S.gaussian_velocities(s, 300.0, seed=41)
host = T._host_run(lambda sy: mdref.RebomosCPU(orc, P, sy, skin=2.0), s, v0, nsteps, every, 2.0, rebuild_every=100)
ctx = capi.Context(0)
p = capi.read_rebomos_file(POT_REBOMOS)
ctx.rebomos_set_params(p)
dev, d = T._device_run(ctx, capi.STYLE_REBOMOS, s, v0, nsteps, every, 2.0, 3.0 * p.rcmax[0][0] + 2.0, [0, 0, 1])
wx, we = T._compare(s, host, dev, xtol=1e-6, etol=1e-7)
out["rebomos"] = dict(atoms=s.n, temp_K=300.0, steps=nsteps, inner_skin=0.5, style_list_builds=int(ctx.md_neighbor_stats()[7]),
                      prunings=ctx.md_prune_stats()["prunings"], worst_dx_A=wx, worst_dE_eV_per_atom=we,
                      etotal_per_atom_host=[host[k][1] / s.n for k in sorted(host)],
                      etotal_per_atom_device=[dev[k][1] / s.n for k in sorted(dev)])
ctx.close()
del os.environ["MDP_INNER_SKIN"]

Tt = orc.aeam_pot(POT_AEAM)
af = capi.AeamFile(POT_AEAM)
s = S.fcc_cell(4.045, 10, frac_type2=0.08, seed=51)
s.mass[1:3] = af.mass[:2]
v0 = S.gaussian_velocities(s, 863.0, seed=52)
host = T._host_run(lambda sy: mdref.AeamCPU(orc, Tt, sy, skin=1.0), s, v0, nsteps, every, 1.0, rebuild_every=25)
ctx = capi.Context(0)
tabs = af.build()
ctx.aeam_set_tables(tabs)
dev, d = T._device_run(ctx, capi.STYLE_AEAM, s, v0, nsteps, every, 1.0, float(af.cut_table(tabs).max()) + 1.0, None)
wx, we = T._compare(s, host, dev, xtol=1e-6, etol=1e-7)
out["aeam"] = dict(atoms=s.n, temp_K=863.0, steps=nsteps, reneighborings=d.builds - 1, prunings=ctx.md_prune_stats()["prunings"],
                   worst_dx_A=wx, worst_dE_eV_per_atom=we,
                   etotal_per_atom_host=[host[k][1] / s.n for k in sorted(host)],
                   etotal_per_atom_device=[dev[k][1] / s.n for k in sorted(dev)])
ctx.close()
print(json.dumps(out, indent=1))
