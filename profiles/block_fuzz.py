"""Randomised parity at scale: random systems of 70 000 - 390 000 atoms (MoS2 replicas scaled by 0.97 - 1.12 with jitter, or
Al-Si alloys with 0 - 20 % Si) run device-resident for 20 - 60 steps from a random temperature (300 - 3 000 K; lists rebuilt
and rows pruned on the device's own triggers), then the forces of ~400-atom blocks at the box corners, the brick seams, the
last tile and random places are compared with the ORACLE's for the same atoms (tests/blockcheck.py).  1e-9 eV/A.
usage: python3 profiles/block_fuzz.py <cases> <seed>"""
import os, sys, random, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests")); sys.path.insert(0, os.path.join(HERE, ".."))
import numpy as np
from conftest import POT_AEAM, POT_REBOMOS
from lammps_plugins_amd.host import capi, resident, system as S
import mdref, oracle_bindings as ob, blockcheck

def main():
    ncase, seed = int(sys.argv[1]), int(sys.argv[2])
    rng = random.Random(seed); orc = ob.load()
    P = orc.rebomos_params(POT_REBOMOS); T = orc.aeam_pot(POT_AEAM)
    rp = capi.read_rebomos_file(POT_REBOMOS); af = capi.AeamFile(POT_AEAM); tabs = af.build()
    bad = 0; t0 = time.time()
    for k in range(ncase):
        style = rng.choice(["rebomos", "aeam"]); sd = rng.randrange(1, 10**6); steps = rng.choice([20, 40, 60])
        try:
            if style == "rebomos":
                rep = rng.choice([(6, 5, 8), (7, 6, 8), (8, 7, 9), (10, 9, 10), (12, 10, 10)])
                fac, amp, temp = rng.choice([0.97, 1.0, 1.0, 1.05, 1.12]), rng.choice([0.0, 0.05, 0.15]), rng.choice([300, 1000, 3000])
                s = S.replicate(S.rebomos_bulk_cell(), rep)
                if fac != 1.0: s = S.scale(s, fac)
                if amp: s = S.jitter(s, amp, seed=sd)
                ctx = capi.Context(0); ctx.rebomos_set_params(rp)
                skin, cutghost, map_, st = 2.0, 3.0 * rp.rcmax[0][0] + 2.0, [0, 0, 1], capi.STYLE_REBOMOS
                if rng.random() < 0.5: os.environ["MDP_INNER_SKIN"] = str(rng.choice([0.4, 0.6, 1.0]))
                else: os.environ.pop("MDP_INNER_SKIN", None)
                factory, shell, margin = (lambda cs: mdref.RebomosCPU(orc, P, cs)), 11.0, 16.0
                desc = f"rep {rep} fac {fac} amp {amp}"
            else:
                n = rng.choice([28, 32, 36, 40, 46]); frac, temp = rng.choice([0.0, 0.0075, 0.08, 0.2]), rng.choice([300, 863, 2500])
                s = S.fcc_cell(4.045, n, frac_type2=frac, seed=sd); s.mass[1:3] = af.mass[:2]
                ctx = capi.Context(0); ctx.aeam_set_tables(tabs)
                skin, cutghost, map_, st = 1.0, float(af.cut_table(tabs).max()) + 1.0, None, capi.STYLE_AEAM
                factory, shell, margin = (lambda cs: mdref.AeamCPU(orc, T, cs)), 13.5, 10.0
                desc = f"cells {n} frac {frac}"
            v0 = S.gaussian_velocities(s, float(temp), seed=sd + 1)
            d = resident.DeviceDomain(ctx, st, s, cutghost, skin, map_, v0=v0)
            d.compute(0, 0)
            for _ in range(steps): d.step(0, 0, rebuild="auto")
            pr = ctx.md_prune_stats()
            got = ctx.md_download(d.nlocal, want=("x", "f"))
            tags, types = d.tags_local, ctx.md_download_int("type", d.nlocal)
            pts = blockcheck.seeds(s.box, got["x"], n_random=3, seed=sd)
            worst, rows = blockcheck.check_blocks(s.box, got["x"], got["f"], types, tags, s.mass, pts, factory, n_interior=400,
                                                  shell=shell, margin=margin, tol=1e-9)
            builds = d.builds
            ctx.close()
            ok, msg = pr["late"] == 0, f"worst dF {worst:.1e} over {len(rows)} blocks prunings {pr['prunings']} late {pr['late']} builds {builds}"
        except AssertionError as e:
            ok, msg = False, f"assert {str(e)[:160]}"
        except Exception as e:  # noqa: BLE001
            ok, msg = False, f"exception {str(e)[-200:]}"
        bad += 0 if ok else 1
        print(f"{'ok ' if ok else 'BAD'} case {k} {style} n {s.n} {desc} T {temp} steps {steps} seed {sd} inner {os.environ.get('MDP_INNER_SKIN')} {msg}", flush=True)
    print(f"{ncase} cases, {bad} bad, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)
main()
