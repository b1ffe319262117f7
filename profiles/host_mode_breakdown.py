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
xa = np.ascontiguousarray(np.concatenate([xw, xw[owner] + shift @ s.box.h.T]))
type_all = np.concatenate([s.type, s.type[owner]]).astype(np.int32)
tag_all = np.concatenate([s.tag, s.tag[owner]]).astype(np.int32)
n = s.n
ctx = capi.Context(0)
ctx.rebomos_set_params(p)
ctx.set_atoms_host(n, xa, type_all, tag_all, 2, map_=[0, 0, 1])
ctx.set_skin(2.0)
f = np.zeros((n, 3))
eng, vir = capi.C.c_double(0.0), np.zeros(6)
def compute():
    ctx._ck(ctx.L.mdp_rebomos_compute_host(ctx.h, 0, 0, capi._dp(f), capi.C.byref(eng), capi._dp(vir), None, None))
compute()
for _ in range(3):
    ctx.set_positions_host(xa); compute()
tu = tc = 0.0
N = 10
ctx.set_timing(True)
km = np.zeros(8)
for _ in range(N):
    t0 = time.perf_counter(); ctx.set_positions_host(xa); t1 = time.perf_counter(); compute(); t2 = time.perf_counter()
    tu += t1 - t0; tc += t2 - t1
    km += np.array(ctx.get_timing())
print(json.dumps({"upload_ms": tu / N * 1e3, "compute_host_ms": tc / N * 1e3, "kernel_phases_ms": (km / N).tolist()[:3]}))
