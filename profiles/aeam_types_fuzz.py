"""Randomised AEAM runs with 3 - 12 atom types (relabelled copies of the bundled two-element file, tests/aeam_five.py: every
new element behaves as Al or as Si, so the oracle on the relabelled file is the reference): tile kernels with per-entry
types for up to 8 types, the generic kernels beyond.  Random type counts, compositions, sizes, temperatures; 30 device-resident
steps, then forces and energy against the oracle on the final positions.  usage: python3 profiles/aeam_types_fuzz.py <cases> <seed>"""
import os, sys, random, time, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests")); sys.path.insert(0, os.path.join(HERE, ".."))
import numpy as np
from conftest import POT_AEAM
from lammps_plugins_amd.host import capi, resident, system as S
import mdref, oracle_bindings as ob, aeam_five

def main():
    ncase, seed = int(sys.argv[1]), int(sys.argv[2])
    rng = random.Random(seed); orc = ob.load(); bad = 0; t0 = time.time()
    tmp = tempfile.mkdtemp()
    for k in range(ncase):
        try:
            nmet, nang = rng.choice([(2, 1), (3, 2), (4, 2), (5, 3), (7, 5), (1, 3), (6, 2)])
            path = os.path.join(tmp, f"p{nmet}_{nang}.aeam")
            if not os.path.exists(path):
                aeam_five.write_relabelled_file(path, POT_AEAM, [0] * nmet + [1] * nang, ["M%d" % i for i in range(nmet)] + ["X%d" % i for i in range(nang)])
            af = capi.AeamFile(path); T = orc.aeam_pot(path); tabs = af.build()
            n = rng.choice([6, 8, 10, 12]); frac = rng.choice([0.01, 0.06, 0.25]); temp = rng.choice([300, 863, 2000]); sd = rng.randrange(1, 10**6)
            s2 = S.jitter(S.fcc_cell(4.045, n, frac_type2=frac, seed=sd), 0.05, seed=sd + 1)
            r = np.random.default_rng(sd)
            ty = np.where(s2.type == 1, r.integers(1, nmet + 1, s2.n), r.integers(nmet + 1, nmet + nang + 1, s2.n)).astype(np.int32)
            s = S.System(s2.box, s2.x.copy(), ty, s2.tag.copy(), np.array([0.0] + list(af.mass)))
            v0 = S.gaussian_velocities(s, float(temp), seed=sd + 2)
            ctx = capi.Context(0); ctx.aeam_set_tables(tabs)
            d = resident.DeviceDomain(ctx, capi.STYLE_AEAM, s, float(af.cut_table(tabs).max()) + 1.0, 1.0, None, v0=v0)
            d.compute(0, 0)
            for _ in range(30): d.step(0, 0, rebuild="auto")
            d.compute(1, 0)
            th = d.thermo(); got = ctx.md_download(d.nlocal, want=("x", "f")); tags = d.tags_local; builds = d.builds
            st = ctx.md_aeam_state(); ctx.close()
            x = np.zeros((s.n, 3)); f = np.zeros((s.n, 3)); x[tags - 1] = got["x"]; f[tags - 1] = got["f"]
            xw = S.wrap(s.box, x)
            o = mdref.AeamCPU(orc, T, S.System(s.box, xw, s.type, s.tag, s.mass)).compute(xw, eflag=1, vflag=0)
            df = float(np.abs(f - o["f_owned"]).max()) / max(1.0, float(np.abs(o["f_owned"]).max())); de = abs(th["pe"] - o["eng"]) / abs(o["eng"])
            ok, msg = df < 1e-9 and de < 1e-10, f"dF {df:.1e} dE {de:.1e} builds {builds}"
        except Exception as e:  # noqa: BLE001
            ok, msg = False, f"exception {type(e).__name__} {str(e)[-200:]}"
        bad += 0 if ok else 1
        print(f"{'ok ' if ok else 'BAD'} case {k} types {nmet}+{nang} cells {n} frac {frac} T {temp} seed {sd} {msg}", flush=True)
    print(f"{ncase} cases, {bad} bad, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)
main()
