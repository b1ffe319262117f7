#!/usr/bin/env python3
"""host-mode step time (PCIe-inclusive, bench.host_mode_rate) of both headline systems with the images kept by the
library (mdp_set_box_host, what the adapters do on one rank) and with MDP_HOST_GHOSTS=upload (positions of all atoms, fp
both ways, ghost forces back: the protocol of rounds 1-3).  usage: python3 profiles/host_mode_ab.py [steps]"""
import json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import bench
import __graft_entry__ as graft
graft.load_package()
from lammps_plugins_amd.host import capi, resident, system as S
E = dict(capi=capi, resident=resident, S=S)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
GOLD = os.path.join(ROOT, "tests", "golden", "potentials")
out = {}
# REBO-MoS 24x24x24 (3.98 M atoms)
p = capi.read_rebomos_file(os.path.join(GOLD, "MoS.REBO.set5b"))
s = S.replicate(S.rebomos_bulk_cell(), (24, 24, 24))
for mode in ("device", "upload"):
    os.environ["MDP_HOST_GHOSTS"] = mode
    ms, img = bench.host_mode_rate(E, s, "rebomos", p, 2.0, 3.0 * p.rcmax[0][0] + 2.0, steps=steps)
    out["rebomos_4m_" + mode] = dict(ms_per_step=round(ms, 3), images_on_device=img)
    print(json.dumps(out), flush=True)
del s
af = capi.AeamFile(os.path.join(GOLD, "AlSi.aeam"))
tabs = af.build()
s = S.jitter(S.fcc_cell(4.045, 63, frac_type2=0.0075, seed=7683797), 0.08, seed=2)
for mode in ("device", "upload"):
    os.environ["MDP_HOST_GHOSTS"] = mode
    ms, img = bench.host_mode_rate(E, s, "aeam", tabs, 1.0, float(af.cut_table(tabs).max()) + 1.0, steps=steps)
    out["aeam_1m_" + mode] = dict(ms_per_step=round(ms, 3), images_on_device=img)
    print(json.dumps(out), flush=True)
