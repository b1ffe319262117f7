#!/usr/bin/env python3
"""where a host-mode REBO-MoS step goes (3.98 M atoms): upload call, compute call (kernels + download + host add), kernel
phases; images kept by the library vs uploaded, alternating.  usage: python3 profiles/host_mode_breakdown.py [rounds]"""
import json, os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from lammps_plugins_amd.host import capi, system as S
pot = os.path.join(ROOT, "tests", "golden", "potentials", "MoS.REBO.set5b")
s = S.replicate(S.rebomos_bulk_cell(), (24, 24, 24))
p = capi.read_rebomos_file(pot)
cutghost = 3.0 * p.rcmax[0][0] + 2.0
xw = S.wrap(s.box, s.x)
owner, shift = S.make_ghosts(s.box, xw, cutghost)
xa = np.ascontiguousarray(np.concatenate([xw, xw[owner] + S.mul_upper(shift, s.box.h)]))
type_all = np.concatenate([s.type, s.type[owner]]).astype(np.int32)
tag_all = np.concatenate([s.tag, s.tag[owner]]).astype(np.int32)
n = s.n
f = np.zeros((n, 3))
eng, vir = capi.C.c_double(0.0), np.zeros(6)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for r in range(rounds):
    for mode in ("upload", "device"):
        os.environ["MDP_HOST_GHOSTS"] = mode
        ctx = capi.Context(0)
        ctx.rebomos_set_params(p)
        ctx.set_box_host(s.box)
        ctx.set_atoms_host(n, xa, type_all, tag_all, 2, map_=[0, 0, 1])
        ctx.set_skin(2.0)

        def compute():
            ctx._ck(ctx.L.mdp_rebomos_compute_host(ctx.h, 0, 0, capi._dp(f), capi.C.byref(eng), capi._dp(vir), None, None))
        compute()
        for _ in range(3):
            ctx.set_positions_host(xa); compute()
        N = 12
        ctx.set_timing(True)
        tu, tc, km = [], [], np.zeros(8)
        for _ in range(N):
            t0 = time.perf_counter(); ctx.set_positions_host(xa); t1 = time.perf_counter(); compute(); t2 = time.perf_counter()
            tu.append(t1 - t0); tc.append(t2 - t1)
            km += np.array(ctx.get_timing())
        print(json.dumps({"mode": mode, "images_on_device": ctx.host_ghosts_derived(), "upload_ms": round(float(np.median(tu)) * 1e3, 3),
                          "compute_host_ms": round(float(np.median(tc)) * 1e3, 3),
                          # mdp_get_timing, rebomos: [0] centre kernels, [1] general kernel, [2] row pruning, [3] LJ tile + cubic
                          "kernel_phases_ms": [round(v, 3) for v in (km / N).tolist()[:4]]}), flush=True)
        ctx.close()
