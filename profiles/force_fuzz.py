"""Randomised force parity: random MoS2 cells (scale 0.92-1.16, jitter up to 0.25 A, small replicas) and random Al-Si alloys
(4-7 fcc cells, 0-50 % Si, jitter up to 0.3 A) through the HIP path (host mode, device-built lists, every tally on, then a
force-only call) against a LIVE oracle compute on the same inputs.  Tolerances of tests/test_gpu_golden.py.
usage: python3 profiles/force_fuzz.py <cases> <seed>"""
import os, sys, random, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests")); sys.path.insert(0, os.path.join(HERE, ".."))
import numpy as np
from conftest import POT_AEAM, POT_REBOMOS
from lammps_plugins_amd.host import capi, system as S
import fixture_cases as FC
import oracle_bindings as ob

def fold(a, owner, n):
    out = a[:n].copy(); np.add.at(out, owner, a[n:]); return out

def main():
    ncase, seed = int(sys.argv[1]), int(sys.argv[2])
    rng = random.Random(seed)
    orc = ob.load()
    P = orc.rebomos_params(POT_REBOMOS); T = orc.aeam_pot(POT_AEAM)
    rctx = capi.Context(0); rp = capi.read_rebomos_file(POT_REBOMOS); rctx.rebomos_set_params(rp)
    af = capi.AeamFile(POT_AEAM); tabs = af.build()
    bad = 0; t0 = time.time()
    for k in range(ncase):
        style = rng.choice(["rebomos", "aeam"])
        try:
            if style == "rebomos":
                fac, amp, sd = rng.uniform(0.92, 1.16), rng.uniform(0.0, 0.25), rng.randrange(10**6)
                rep = rng.choice([None, (2, 1, 1), (1, 2, 1), (2, 2, 1), (1, 1, 2)])
                s = FC._rebomos(fac, amp, sd, rep); desc = f"fac {fac:.3f} amp {amp:.3f} seed {sd} rep {rep}"
                eng = FC.engine(style, s, orc, P=P); want = FC.oracle_outputs(style, eng, s.x)
                xa = eng.all_positions(s.x)
                lists = rng.choice(["device", "host_csr"]); desc += f" lists {lists}"
                rctx.set_atoms_host(eng.nlocal, xa, eng.type_all, eng.tag_all, 2, map_=[0, 0, 1])
                if lists == "device": rctx.set_skin(2.0)
                else: rctx.set_neighbors_csr_host(eng.nn, eng.off, eng.nb, 2.0)
                g = rctx.rebomos_compute_host(eng.nlocal, eflag=3, vflag=5)
                g0 = rctx.rebomos_compute_host(eng.nlocal, eflag=0, vflag=0)
                fs = max(1.0, float(np.abs(want["f"]).max()))
                err = dict(f=np.abs(g["f"] - want["f"]).max() / fs, f0=np.abs(g0["f"] - want["f"]).max() / fs,
                           e=abs(g["eng"] - float(want["eng"])) / abs(float(want["eng"])), ea=np.abs(g["eatom"] - want["eatom"]).max(),
                           va=np.abs(g["vatom"] - want["vatom"]).max() / max(1.0, np.abs(want["vatom"]).max()))
                lim = dict(f=1e-9, f0=1e-9, e=1e-10, ea=1e-9, va=1e-9)
            else:
                nc, frac, amp, sd = rng.choice([4, 5, 6, 7]), rng.choice([0.0, 0.0075, 0.03, 0.08, 0.2, 0.5]), rng.uniform(0.0, 0.3), rng.randrange(10**6)
                s = FC._aeam_cell(nc, frac, amp, sd); desc = f"cells {nc} frac {frac} amp {amp:.3f} seed {sd}"
                if rng.random() < 0.5:   # a sheared (triclinic) box: same lamda coordinates in a tilted cell
                    L = float(s.box.prd[0]); tilt = np.array([rng.uniform(-0.12, 0.12) * L for _ in range(3)])
                    nb = S.Box(s.box.lo.copy(), s.box.prd.copy(), tilt)
                    s = S.System(nb, nb.lamda2x(s.box.x2lamda(s.x)), s.type, s.tag, s.mass); desc += f" tilt {np.round(tilt, 2).tolist()}"
                eng = FC.engine(style, s, orc, T=T); want = FC.oracle_outputs(style, eng, s.x)
                lists = rng.choice(["device", "host_csr"]); desc += f" lists {lists}"
                ctx = capi.Context(0); ctx.aeam_set_tables(tabs)
                if lists == "device":
                    cut = float(af.cut_table(tabs).max()) + 1.0
                    xa, type_all, tag_all, owner, _, nloc, _ = S.with_ghosts(s, cut)
                    ctx.aeam_device_lists(True)
                    ctx.set_atoms_host(nloc, xa, type_all, tag_all, 2, map_=None); ctx.set_skin(1.0)
                else:
                    xa, owner, nloc = eng.all_positions(s.x), eng.owner, eng.nlocal
                    ctx.set_atoms_host(nloc, xa, eng.type_all, eng.tag_all, 2, map_=None)
                    ctx.set_neighbors_csr_host(eng.nn, eng.off, eng.nb, 1.0)
                d = ctx.aeam_density_host(nloc, eflag=3)
                r = ctx.aeam_force_host(len(xa), nloc, np.concatenate([d["fp"], d["fp"][owner]]), eflag=3, vflag=5)
                d0 = ctx.aeam_density_host(nloc, eflag=0)
                r0 = ctx.aeam_force_host(len(xa), nloc, np.concatenate([d0["fp"], d0["fp"][owner]]), eflag=0, vflag=0)
                ctx.close()
                fs = max(1.0, float(np.abs(want["f"]).max()))
                err = dict(f=np.abs(ob.fold_ghost_forces(r["f"], owner, nloc) - want["f"]).max() / fs,
                           f0=np.abs(ob.fold_ghost_forces(r0["f"], owner, nloc) - want["f"]).max() / fs,
                           e=abs(d["eng"] + r["eng"] - float(want["eng"])) / abs(float(want["eng"])),
                           ea=np.abs(d["eatom"] + r["eatom"] - want["eatom"]).max(),
                           va=np.abs(fold(r["vatom"], owner, nloc) - want["vatom"]).max() / max(1.0, np.abs(want["vatom"]).max()),
                           rho=np.abs(d["rho"] - want["rho"]).max() / max(1.0, np.abs(want["rho"]).max()))
                lim = dict(f=1e-9, f0=1e-9, e=1e-10, ea=1e-9, va=1e-9, rho=1e-11)
            ok = all(err[q] < lim[q] for q in lim)
        except Exception as e:  # noqa: BLE001
            ok, err, desc = False, {"exception": str(e)[-300:]}, desc if "desc" in dir() else "?"
        bad += 0 if ok else 1
        print(f"{'ok ' if ok else 'BAD'} case {k} {style} {desc} n {s.n} " + " ".join(f"{q} {v:.1e}" if isinstance(v, float) or hasattr(v, 'dtype') else f"{q} {v}" for q, v in err.items()), flush=True)
    print(f"{ncase} cases, {bad} bad, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)
main()
