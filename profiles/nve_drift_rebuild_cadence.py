import json, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from lammps_plugins_amd.host import capi, resident, system as S
s = S.replicate(S.rebomos_bulk_cell(), (12, 12, 12))
p = capi.read_rebomos_file(os.path.join(ROOT, "tests", "golden", "potentials", "MoS.REBO.set5b"))
v0 = S.gaussian_velocities(s, 900.0, seed=12345)
res = {}
for every in (0, 20):
    ctx = capi.Context(0)
    ctx.rebomos_set_params(p)
    d = resident.DeviceDomain(ctx, capi.STYLE_REBOMOS, s, 3.0 * p.rcmax[0][0] + 2.0, 2.0, [0, 0, 1], v0=v0)
    d.compute(1, 0)
    t = d.thermo(); e0 = t["pe"] + t["ke"]
    out = []
    for step in range(1, 3001):
        rb = (step % every == 0) if every else "auto"
        d.step(1 if step % 500 == 0 else 0, 0, rebuild=rb)
        if step % 500 == 0:
            t = d.thermo(); out.append(round((t["pe"] + t["ke"] - e0) / s.n, 8))
    res["forced every %d" % every if every else "auto"] = dict(builds=d.builds, drift=out)
    ctx.close()
print(json.dumps(res))
