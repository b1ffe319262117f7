"""Randomised runs of what `fix nve/mdp` does on one rank in its default mode (the C-ABI underneath: mdp_hnve_initial / compute
with f == NULL / mdp_hnve_final, the images kept by the library, the device's displacement check read one step late and a
HOST reneighboring -- download, wrap, new ghosts, re-upload of atoms and velocities -- when it fires): random MoS2 cells, random
temperatures up to 3 000 K, a projectile in some, against velocity-Verlet on the host around the ORACLE (the loop of
tests/test_gpu_trajectory.py).  usage: python3 profiles/hnve_fuzz.py <cases> <seed>"""
import os, sys, random, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests")); sys.path.insert(0, os.path.join(HERE, ".."))
import numpy as np
from conftest import POT_REBOMOS
from lammps_plugins_amd.host import capi, system as S
import mdref, oracle_bindings as ob
import test_gpu_trajectory as TT

def main():
    ncase, seed = int(sys.argv[1]), int(sys.argv[2])
    rng = random.Random(seed); orc = ob.load(); P = orc.rebomos_params(POT_REBOMOS)
    bad = 0; t0 = time.time()
    for k in range(ncase):
        sd = rng.randrange(1, 10**6); nsteps = 90
        rep = rng.choice([None, (2, 1, 1), (1, 2, 1)]); temp = rng.choice([300, 1200, 3000]); skin = rng.choice([1.0, 2.0])
        try:
            s = S.rebomos_bulk_cell() if rep is None else S.replicate(S.rebomos_bulk_cell(), rep)
            v0 = S.gaussian_velocities(s, float(temp), seed=sd)
            shot = rng.random() < 0.4
            if shot: v0[rng.randrange(s.n)] += np.array([20.0, -22.0, 18.0])
            if rng.random() < 0.5: os.environ["MDP_INNER_SKIN"] = str(rng.choice([0.3, 0.5]))
            else: os.environ.pop("MDP_INNER_SKIN", None)
            host = TT._host_run(lambda sy: mdref.RebomosCPU(orc, P, sy, skin=skin), s, v0, nsteps, nsteps, skin, rebuild_every=10)
            c = capi.Context(0); c.rebomos_set_params(ob.product_rebomos_params(P))
            x = S.wrap(s.box, s.x); v = v0.copy(); rebuilds = 0

            def upload(x, v):
                eng = mdref.RebomosCPU(None, P, S.System(s.box, x.copy(), s.type, s.tag, s.mass), skin=skin)   # ghosts only
                c.set_box_host(s.box)
                c.set_atoms_host(eng.nlocal, eng.all_positions(x), eng.type_all, eng.tag_all, 2, map_=[0, 0, 1])
                c.set_skin(skin)
                c.hnve_upload_v(v)
            c.hnve_setup(0.001, S.FTM2V, s.mass)
            upload(x, v)
            c.rebomos_compute_host(s.n, eflag=0, vflag=0)
            late_any = False
            for step in range(nsteps):
                moved, late = c.hnve_initial()
                late_any |= late
                if moved:   # the host reneighbors: what Verlet::run does when Neighbor::decide() says so
                    got = c.hnve_download(s.n, want=("x", "v"))
                    upload(S.wrap(s.box, got["x"]), got["v"]); rebuilds += 1
                c._ck(c.L.mdp_rebomos_compute_host(c.h, 0, 0, None, None, None, None, None))
                c.hnve_final()
            got = c.hnve_download(s.n, want=("x", "v")); c.close()
            xh = host[nsteps][0]
            dx = got["x"] - xh; dx -= np.round(s.box.x2lamda(dx + s.box.lo)) @ s.box.h.T
            ex = float(np.abs(dx).max())
            ok = ex < (1e-6 if (temp >= 3000 or shot) else 1e-8) and not late_any
            msg = f"dx {ex:.1e} host rebuilds {rebuilds} late {late_any}"
        except Exception as e:  # noqa: BLE001
            ok, msg = False, f"exception {type(e).__name__} {str(e)[-200:]}"
        bad += 0 if ok else 1
        print(f"{'ok ' if ok else 'BAD'} case {k} n {s.n} T {temp} skin {skin} shot {shot} seed {sd} inner {os.environ.get('MDP_INNER_SKIN')} {msg}", flush=True)
    print(f"{ncase} cases, {bad} bad, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)
main()
