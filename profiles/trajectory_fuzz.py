"""Randomised device-resident trajectories against the same trajectories driven on the host with ORACLE forces (the harness of
tests/test_gpu_trajectory.py): random small MoS2 cells and Al-Si alloys, random temperatures up to 4 000 K (one atom in 64
would not do here), random seeds, a fast projectile in some of them, 120 steps with the device's own deferred checks, row
prunings, list rebuilds and reneighborings.  Positions 1e-8 A (hot cases 1e-6: the trajectories are chaotic), energy 1e-9 eV
per atom.  usage: python3 profiles/trajectory_fuzz.py <cases> <seed>"""
import os, sys, random, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests")); sys.path.insert(0, os.path.join(HERE, ".."))
import numpy as np
from conftest import POT_AEAM, POT_REBOMOS
from lammps_plugins_amd.host import capi, system as S
import mdref, oracle_bindings as ob
import test_gpu_trajectory as TT

def main():
    ncase, seed = int(sys.argv[1]), int(sys.argv[2])
    rng = random.Random(seed); orc = ob.load()
    P = orc.rebomos_params(POT_REBOMOS); T = orc.aeam_pot(POT_AEAM)
    rp = capi.read_rebomos_file(POT_REBOMOS); af = capi.AeamFile(POT_AEAM); tabs = af.build()
    bad = 0; t0 = time.time()
    for k in range(ncase):
        style = rng.choice(["rebomos", "aeam"]); sd = rng.randrange(1, 10**6); nsteps = 120
        try:
            if style == "rebomos":
                rep = rng.choice([None, (2, 1, 1), (1, 2, 1)]); temp = rng.choice([300, 1500, 4000]); skin = 2.0
                s = S.rebomos_bulk_cell() if rep is None else S.replicate(S.rebomos_bulk_cell(), rep)
                if rng.random() < 0.5: os.environ["MDP_INNER_SKIN"] = str(rng.choice([0.3, 0.5, 1.0]))
                else: os.environ.pop("MDP_INNER_SKIN", None)
            else:
                n = rng.choice([4, 5, 6]); temp = rng.choice([300, 863, 2500]); skin = 1.0
                s = S.fcc_cell(4.045, n, frac_type2=rng.choice([0.0, 0.03, 0.2]), seed=sd); s.mass[1:3] = af.mass[:2]
                if rng.random() < 0.5:   # a sheared (triclinic) box
                    L = float(s.box.prd[0]); tilt = np.array([rng.uniform(-0.06, 0.06) * L for _ in range(3)])
                    nb = S.Box(s.box.lo.copy(), s.box.prd.copy(), tilt)
                    s = S.System(nb, nb.lamda2x(s.box.x2lamda(s.x)), s.type, s.tag, s.mass)
            v0 = S.gaussian_velocities(s, float(temp), seed=sd + 1)
            shot = rng.random() < 0.4
            if shot: v0[rng.randrange(s.n)] += np.array([rng.choice([-1, 1]) * 22.0, 20.0, 18.0])   # a projectile at ~35 A/ps
            if style == "rebomos":
                host = TT._host_run(lambda sy: mdref.RebomosCPU(orc, P, sy, skin=skin), s, v0, nsteps, 30, skin, rebuild_every=10)
                ctx = capi.Context(0); ctx.rebomos_set_params(rp)
                dev, d = TT._device_run(ctx, capi.STYLE_REBOMOS, s, v0, nsteps, 30, skin, 3.0 * rp.rcmax[0][0] + skin, [0, 0, 1])
            else:
                host = TT._host_run(lambda sy: mdref.AeamCPU(orc, T, sy, skin=skin), s, v0, nsteps, 30, skin, rebuild_every=5)
                ctx = capi.Context(0); ctx.aeam_set_tables(tabs)
                dev, d = TT._device_run(ctx, capi.STYLE_AEAM, s, v0, nsteps, 30, skin, float(af.cut_table(tabs).max()) + skin, None)
            pr = ctx.md_prune_stats(); ctx.close()
            hot = temp >= 2500 or shot
            wx, we = TT._compare(s, host, dev, xtol=1e-6 if hot else 1e-8, etol=1e-8 if hot else 1e-9)
            ok, msg = True, f"dx {wx:.1e} dE/atom {we:.1e} prunings {pr['prunings']} late {pr['late']} builds {d.builds}"
            if pr["late"]: ok, msg = False, msg + " LATE PRUNING"
        except AssertionError as e:
            ok, msg = False, f"assert {str(e)[:120]}"
        except Exception as e:  # noqa: BLE001
            ok, msg = False, f"exception {str(e)[-200:]}"
        bad += 0 if ok else 1
        print(f"{'ok ' if ok else 'BAD'} case {k} {style} n {s.n} tilt {np.round(s.box.tilt, 1).tolist()} T {temp} shot {shot} seed {sd} inner {os.environ.get('MDP_INNER_SKIN')} {msg}", flush=True)
    print(f"{ncase} cases, {bad} bad, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)
main()
