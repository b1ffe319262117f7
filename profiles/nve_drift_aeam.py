"""NVE consistency of the aeam path: 256 000 atoms (0.75 % Si) at 863 K, 3000 steps of 1 fs with the displacement check
every step, once with the persistent density kernel (spline table in LDS) and once with the gather kernel.  Prints
(E_total - E_0) per atom every 500 steps and the number of reneighborings."""
import json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from lammps_plugins_amd.host import capi, resident, system as S
af = capi.AeamFile(os.path.join(ROOT, "tests", "golden", "potentials", "AlSi.aeam"))
tabs = af.build()
s = S.fcc_cell(4.045, 40, frac_type2=0.0075, seed=7683797)
s.mass[1:3] = af.mass
v0 = S.gaussian_velocities(s, 863.0, seed=4928459)
cutghost = float(af.cut_table(tabs).max()) + 1.0
res = {}
for tag, env, prune in (("persistent density kernel", "1", "1"), ("gather kernels", "0", "1"), ("persistent density kernel, rows as built", "1", "0")):
    os.environ["MDP_AEAM_PERSIST"] = env
    os.environ["MDP_PRUNE"] = prune
    ctx = capi.Context(0)
    ctx.aeam_set_tables(tabs)
    d = resident.DeviceDomain(ctx, capi.STYLE_AEAM, s, cutghost, 1.0, None, v0=v0)
    d.compute(1, 0)
    t = d.thermo(); e0 = t["pe"] + t["ke"]
    out = []
    for step in range(1, 3001):
        d.step(1 if step % 500 == 0 else 0, 0, rebuild="auto")
        if step % 500 == 0:
            t = d.thermo(); out.append(round((t["pe"] + t["ke"] - e0) / s.n, 9))
    res[tag] = dict(atoms=s.n, builds=d.builds, prune=ctx.md_prune_stats(), e0_per_atom=round(e0 / s.n, 9), drift_eV_per_atom=out,
                    temp_end=round(S.temperature(t["ke"], s.n), 2))
    ctx.close()
print(json.dumps(res))
