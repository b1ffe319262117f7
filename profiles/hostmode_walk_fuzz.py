"""Randomised HOST-MODE walks (the plain plugin path: atoms uploaded once, then positions every step -- mdp_set_positions_host
-- with the library deciding by itself when its own lists and its pruned rows are stale): random cells, then 40 steps of a
random walk (every atom a little, one atom a lot, some steps nobody) inside the host's skin, forces against the oracle at
EVERY step.  usage: python3 profiles/hostmode_walk_fuzz.py <cases> <seed>"""
import os, sys, random, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests")); sys.path.insert(0, os.path.join(HERE, ".."))
import numpy as np
from conftest import POT_AEAM, POT_REBOMOS
from lammps_plugins_amd.host import capi, system as S
import fixture_cases as FC
import oracle_bindings as ob

def main():
    ncase, seed = int(sys.argv[1]), int(sys.argv[2])
    rng = random.Random(seed); orc = ob.load()
    P = orc.rebomos_params(POT_REBOMOS); T = orc.aeam_pot(POT_AEAM)
    rp = capi.read_rebomos_file(POT_REBOMOS); af = capi.AeamFile(POT_AEAM); tabs = af.build()
    bad = 0; t0 = time.time()
    for k in range(ncase):
        style = rng.choice(["rebomos", "aeam"]); sd = rng.randrange(1, 10**6); nr = np.random.default_rng(sd)
        worst = 0.0; msg = ""
        images = rng.random() < 0.5     # one periodic rank: the library keeps the images itself (mdp_set_box_host), the host's ghost positions are poison
        try:
            if style == "rebomos":
                s = FC._rebomos(rng.uniform(0.95, 1.12), rng.uniform(0.0, 0.15), sd, rng.choice([None, (2, 1, 1), (1, 2, 1)])); skin = 2.0
                eng = FC.engine(style, s, orc, P=P)
                ctx = capi.Context(0); ctx.rebomos_set_params(rp)
                if images: ctx.set_box_host(s.box)
                ctx.set_atoms_host(eng.nlocal, eng.all_positions(s.x), eng.type_all, eng.tag_all, 2, map_=[0, 0, 1]); ctx.set_skin(skin)
                if rng.random() < 0.5: os.environ["MDP_INNER_SKIN"] = str(rng.choice([0.3, 0.6]))
                else: os.environ.pop("MDP_INNER_SKIN", None)
            else:
                s = FC._aeam_cell(rng.choice([4, 5, 6]), rng.choice([0.0075, 0.08, 0.3]), rng.uniform(0.0, 0.15), sd); skin = 1.0
                eng = FC.engine(style, s, orc, T=T)
                ctx = capi.Context(0); ctx.aeam_set_tables(tabs); ctx.aeam_device_lists(True)
                if images: ctx.set_box_host(s.box)
                ctx.set_atoms_host(eng.nlocal, eng.all_positions(s.x), eng.type_all, eng.tag_all, 2, map_=None); ctx.set_skin(skin)
            x = s.x.copy(); x0 = x.copy(); n = s.n; owner = eng.owner
            for step in range(40):
                mode = rng.choice(["all", "all", "one", "none", "few"])
                dxs = np.zeros((n, 3))
                if mode == "all": dxs = nr.normal(0.0, rng.choice([0.002, 0.01, 0.03]), (n, 3))
                elif mode == "one": dxs[rng.randrange(n)] = nr.normal(0.0, 0.15, 3)
                elif mode == "few": dxs[nr.integers(0, n, 5)] = nr.normal(0.0, 0.08, (5, 3))
                xn = x + dxs
                # stay inside the host's list: no atom further than 0.45 skin from where the lists were built
                far = np.linalg.norm(xn - x0, axis=1) > 0.45 * skin
                xn[far] = x[far]
                x = xn
                xa = eng.all_positions(x)
                if images:
                    xa = xa.copy(); xa[eng.nlocal:] = np.nan
                ctx.set_positions_host(xa)
                o = eng.compute(x, eflag=1, vflag=0)
                if style == "rebomos":
                    g = ctx.rebomos_compute_host(eng.nlocal, eflag=0 if step % 3 else 1, vflag=0)
                    f = g["f"]
                else:
                    if images:   # fp and the images' share of the three-body forces stay on the device
                        d = ctx.aeam_density_host(eng.nlocal, eflag=0, keep_fp=True)
                        r = ctx.aeam_force_host(len(xa), eng.nlocal, None, eflag=0, vflag=0)
                        f = r["f"][:eng.nlocal]
                    else:
                        d = ctx.aeam_density_host(eng.nlocal, eflag=0)
                        r = ctx.aeam_force_host(len(xa), eng.nlocal, np.concatenate([d["fp"], d["fp"][owner]]), eflag=0, vflag=0)
                        f = ob.fold_ghost_forces(r["f"], owner, eng.nlocal)
                err = float(np.abs(f - o["f_owned"]).max()) / max(1.0, float(np.abs(o["f_owned"]).max()))
                if err > worst: worst, msg = err, f"worst at step {step} ({mode})"
            ctx.close()
            ok = worst < 1e-9
        except Exception as e:  # noqa: BLE001
            ok, msg = False, f"exception {type(e).__name__} {str(e)[-200:]}"
        bad += 0 if ok else 1
        print(f"{'ok ' if ok else 'BAD'} case {k} {style} n {s.n} images {images} seed {sd} inner {os.environ.get('MDP_INNER_SKIN')} dF {worst:.1e} {msg}", flush=True)
    print(f"{ncase} cases, {bad} bad, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)
main()
