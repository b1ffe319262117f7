"""Randomised check of the dynamic row pruning and of the pair queues (one GPU, minihost/ddhost.cpp): random hot / drifting /
strained-by-temperature runs with the kernels walking PRUNED rows (default) against the same run with MDP_PRUNE=0 (rows as
built), against MDP_LJ_QUEUE=1 / 0 (cubic-branch pairs queued / found by a second walk) and against other inner skins of the
style's own lists (MDP_INNER_SKIN: other rebuild steps, the same pairs).  The validity of pruned rows
rests on a displacement trigger read one step late with a margin: a pair missed because of it would show here as a
trajectory that leaves its twin.  usage: python3 profiles/prune_fuzz.py <cases> <seed>"""
import os, sys, random, tempfile, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests")); sys.path.insert(0, os.path.join(HERE, ".."))
import numpy as np
import test_gpu_ddhost as T

def run(args, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return T._ddhost(args)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v

def main():
    ncase, seed = int(sys.argv[1]), int(sys.argv[2])
    rng = random.Random(seed); bad = 0; t0 = time.time()
    for k in range(ncase):
        style = rng.choice(["rebomos", "aeam"])
        if style == "rebomos":
            rep = rng.choice([(3, 3, 2), (4, 4, 2), (5, 3, 2)]); temp = rng.choice([300, 1500, 3000, 5000]); extra = []
        else:
            n = rng.choice([12, 16, 20]); rep = (n, n, n); temp = rng.choice([300, 863, 2000]); extra = ["-frac2", rng.choice([0.0075, 0.08])]
        drift = [rng.choice([-60, 0, 40, 90]) for _ in range(3)]
        steps = rng.choice([80, 150, 250]); sd = rng.randrange(1, 10**7)
        common = ["-style", style, "-ranks", 1, "-replicate", *rep, "-steps", steps, "-thermo", steps, "-temp", temp, "-seed", sd, "-drift", *drift] + extra
        res = {}
        with tempfile.TemporaryDirectory() as d:
            try:
                for name, env in (("pruned", {}), ("as_built", {"MDP_PRUNE": "0"}), ("queued", {"MDP_LJ_QUEUE": "1"}), ("walked", {"MDP_LJ_QUEUE": "0"}),
                                  ("inner_skin_0.3", {"MDP_INNER_SKIN": "0.3"}), ("inner_skin_1.2", {"MDP_INNER_SKIN": "1.2"})):
                    if style == "aeam" and name not in ("pruned", "as_built"): continue
                    out, _ = run(common + ["-dump", os.path.join(d, name)], env)
                    res[name] = T._dump(os.path.join(d, name), 1) + (out.split("Neighbor list builds = ")[1].split()[0],)
                ref = res["as_built"]
                errs = {n: (float(np.abs(r[0] - ref[0]).max()), float(np.abs(r[1] - ref[1]).max())) for n, r in res.items() if n != "as_built"}
                ok = all(e[0] < 1e-9 and e[1] < 1e-8 for e in errs.values())
            except Exception as e:  # noqa: BLE001
                ok, errs = False, {"exception": str(e)[-200:]}
        bad += 0 if ok else 1
        print(f"{'ok ' if ok else 'BAD'} case {k} {style} rep {rep} T {temp} drift {drift} steps {steps} seed {sd} {extra} builds {res.get('pruned', (0, 0, '?'))[2]} " +
              " ".join(f"{n} dx {e[0]:.1e} dv {e[1]:.1e}" if isinstance(e, tuple) else f"{n} {e}" for n, e in errs.items()), flush=True)
    print(f"{ncase} cases, {bad} bad, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)
main()
