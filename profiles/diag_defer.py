"""diagnostic: where do the deferred-final and the separate-kernel REBO-MoS runs part (tests/test_gpu_resident.py)?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
graft.load_package()
from lammps_plugins_amd.host import capi, resident, system as S
POT = os.path.join(ROOT, "tests", "golden", "potentials", "MoS.REBO.set5b")

def run(defer, steps=40):
    ctx = capi.Context(0)
    p = capi.read_rebomos_file(POT)
    ctx.rebomos_set_params(p)
    s = S.replicate(S.rebomos_bulk_cell(), (3, 3, 3))
    v0 = S.gaussian_velocities(s, 900.0, seed=3)
    d = resident.DeviceDomain(ctx, capi.STYLE_REBOMOS, s, 3.0 * p.rcmax[0][0] + 2.0, 2.0, [0, 0, 1], v0=v0)
    d.compute(0, 0)
    out = []
    for k in range(1, steps + 1):
        ev = 1 if k % 10 == 0 else 0
        d.step(ev, 0, rebuild="auto", defer_final=defer and not ev)
        if ev:
            d.thermo()
        d.flush()
        got = ctx.md_download(d.nlocal, want=("x", "f"))
        order = np.argsort(d.tags_local)
        st = ctx.md_list_state()
        out.append((got["x"][order], got["f"][order], d.builds, ctx.rebomos_list_info()["builds"], ctx.md_prune_stats()["prunings"], st))
    ctx.close()
    return out

a, b, c = run(False), run(False), run(True)
for k in range(len(a)):
    print(k + 1, "same-mode dx", np.abs(a[k][0] - b[k][0]).max(), "df", np.abs(a[k][1] - b[k][1]).max(),
          "| defer dx", np.abs(a[k][0] - c[k][0]).max(), "df", np.abs(a[k][1] - c[k][1]).max(),
          "| builds", a[k][2], c[k][2], "style", a[k][3], b[k][3], c[k][3], "prunes", a[k][4], b[k][4], c[k][4], "ovf", a[k][5]["centre3_overflow"], b[k][5]["centre3_overflow"], c[k][5]["centre3_overflow"], a[k][5]["centre3_list_mode"])
