#!/usr/bin/env python3
"""where a host-mode AEAM step goes (1.0 M atoms, device lists): upload call, density half, force half.
usage: python3 profiles/host_mode_breakdown_aeam.py [rounds]"""
import json, os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from lammps_plugins_amd.host import capi, system as S
af = capi.AeamFile(os.path.join(ROOT, "tests", "golden", "potentials", "AlSi.aeam"))
tabs = af.build()
s = S.jitter(S.fcc_cell(4.045, 63, frac_type2=0.0075, seed=7683797), 0.08, seed=2)
xw = S.wrap(s.box, s.x)
owner, shift = S.make_ghosts(s.box, xw, float(af.cut_table(tabs).max()) + 1.0)
xa = np.ascontiguousarray(np.concatenate([xw, xw[owner] + S.mul_upper(shift, s.box.h)]))
type_all = np.concatenate([s.type, s.type[owner]]).astype(np.int32)
tag_all = np.concatenate([s.tag, s.tag[owner]]).astype(np.int32)
n, nall = s.n, len(xa)
f = np.zeros((nall, 3)); fp = np.zeros(nall)
eng, vir = capi.C.c_double(0.0), np.zeros(6)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for r in range(rounds):
    for mode in ("upload", "device"):
        os.environ["MDP_HOST_GHOSTS"] = mode
        ctx = capi.Context(0)
        ctx.aeam_set_tables(tabs)
        ctx.aeam_device_lists(True)
        ctx.set_box_host(s.box)
        ctx.set_atoms_host(n, xa, type_all, tag_all, 2)
        ctx.set_skin(1.0)
        keep = ctx.host_ghosts_derived()

        def dens():
            ctx._ck(ctx.L.mdp_aeam_density_host(ctx.h, 0, None if keep else capi._dp(fp), None, capi.C.byref(eng), None))
            if not keep:
                fp[n:] = fp[owner]

        def force():
            ctx._ck(ctx.L.mdp_aeam_force_host(ctx.h, 0, 0, None if keep else capi._dp(fp), capi._dp(f), capi.C.byref(eng),
                                              capi._dp(vir), None, None))
        dens(); force()
        for _ in range(3):
            ctx.set_positions_host(xa); dens(); force()
        tu, td, tf = [], [], []
        for _ in range(15):
            t0 = time.perf_counter(); ctx.set_positions_host(xa); t1 = time.perf_counter(); dens(); t2 = time.perf_counter(); force(); t3 = time.perf_counter()
            tu.append(t1 - t0); td.append(t2 - t1); tf.append(t3 - t2)
        med = lambda v: round(float(np.median(v)) * 1e3, 3)
        print(json.dumps({"mode": mode, "images_on_device": keep, "upload_ms": med(tu), "density_half_ms": med(td), "force_half_ms": med(tf),
                          "step_ms": med(np.array(tu) + np.array(td) + np.array(tf))}), flush=True)
        ctx.close()
