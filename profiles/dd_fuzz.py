"""Randomised multi-rank consistency runs of the C++ resident host (minihost/ddhost.cpp) on one GPU through the RCCL test
double: for random styles, rank counts, sizes, temperatures, drifts and seeds the N-rank run must end where the one-rank
run ends (positions 1e-8 A, velocities 1e-7 A/ps).  usage: python3 profiles/dd_fuzz.py <cases> <seed> [style]
(The scan bug of DESIGN section 6 showed on one configuration in dozens: this is the net for its kind.)"""
import os, sys, random, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import test_gpu_ddhost as T
from lammps_plugins_amd.host import system as S

def main():
    ncase, seed = int(sys.argv[1]), int(sys.argv[2])
    only = sys.argv[3] if len(sys.argv) > 3 else None
    rng = random.Random(seed)
    bad = 0
    t0 = time.time()
    for k in range(ncase):
        style = only or rng.choice(["rebomos", "aeam"])
        ranks = rng.choice([2, 3, 4, 6, 8])
        if style == "rebomos":
            rep = rng.choice([(3, 3, 2), (4, 2, 2), (2, 4, 3), (5, 3, 2), (3, 2, 4)])
            temp = rng.choice([300, 900, 1500, 3000, 5000])
            box = S.replicate(S.rebomos_bulk_cell(), rep).box
            extra = []
        else:
            n = rng.choice([12, 14, 16, 18])
            rep = (n, n, n)
            temp = rng.choice([300, 863, 1200, 3000])
            box = S.fcc_cell(4.045, n).box
            extra = ["-frac2", rng.choice([0.0, 0.0075, 0.03, 0.08])]
        drift = [rng.choice([-60, -30, 0, 25, 40, 70]) for _ in range(3)]
        steps = rng.choice([40, 60, 90])
        sd = rng.randrange(1, 10**7)
        thermo = rng.choice([steps, 10, 7])     # (energy / virial steps in mid-run: the other kernel variants, sums over ranks)
        common = ["-style", style, "-replicate", *rep, "-steps", steps, "-thermo", thermo, "-temp", temp, "-seed", sd, "-drift", *drift] + extra
        with tempfile.TemporaryDirectory() as d:
            try:
                _, rows1 = T._ddhost(["-ranks", 1, "-dump", os.path.join(d, "one")] + common)
                out, rowsn = T._ddhost(["-ranks", ranks, "-dump", os.path.join(d, "many")] + common, double=True)
                x1, v1 = T._dump(os.path.join(d, "one"), 1)
                xn, vn = T._dump(os.path.join(d, "many"), ranks)
                dx = xn - x1
                dx -= np.round(box.x2lamda(dx + box.lo)) @ box.h.T
                ex, ev = float(np.abs(dx).max()), float(np.abs(vn - v1).max())
                ok = ex < 1e-8 and ev < 1e-7 and len(rows1) == len(rowsn) and len(rows1) >= 2
                for a, b in zip(rowsn, rows1):       # step temp press pe ke as printed (%.8g or better)
                    ok = ok and all(abs(u - v) <= 2e-7 * max(abs(v), 1.0) for u, v in zip(a, b))
                builds = out.split("Neighbor list builds = ")[1].split()[0]
            except Exception as e:  # noqa: BLE001
                ok, ex, ev, builds = False, -1, -1, str(e)[-200:]
        bad += 0 if ok else 1
        print(f"{'ok ' if ok else 'BAD'} case {k} {style} ranks {ranks} rep {rep} T {temp} drift {drift} steps {steps} seed {sd} {extra} dx {ex:.2e} dv {ev:.2e} builds {builds}", flush=True)
    print(f"{ncase} cases, {bad} bad, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)

main()
