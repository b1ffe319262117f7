"""Randomised runs of the plugins on N ranks of the mini-host: for random MoS2 / Al-Si inputs (size, temperature, skin, run
length, seed) the thermo rows of `minilmp -np N` -- in host mode and under `fix nve/mdp` on the library's bricks; on one rank: the fix in its
default mode and with `bricks yes` -- must be the rows of the one-rank run with the host's own `fix nve` (printed digits: rel 5e-7).  usage: python3 profiles/minilmp_fuzz.py <cases> <seed>"""
import os, sys, random, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests")); sys.path.insert(0, os.path.join(HERE, ".."))
from test_plugin_boundary import PKG, _run, _thermo_rows
from lammps_plugins_amd.host import capi

def rows_equal(a, b):
    if len(a) != len(b) or not a: return False
    for r, q in zip(a, b):
        for u, v in zip(r, q):
            if abs(u - v) > 5e-7 * max(abs(v), 1.0) + 1e-5: return False
    return True

def main():
    ncase, seed = int(sys.argv[1]), int(sys.argv[2])
    rng = random.Random(seed)
    env = dict(MDP_RCCL_LIBRARY=capi.FAKE_RCCL, MDP_FAKE_RCCL_TIMEOUT_S="60", MDP_FIX_STATS="1")
    reb = open(os.path.join(PKG, "examples", "in.rebomos-bulk.mi355x")).read()
    aea = open(os.path.join(PKG, "examples", "in.aeam-alsi.mi355x")).read()
    bad = 0; t0 = time.time()
    for k in range(ncase):
        np_ = rng.choice([1, 1, 2, 3, 4, 6, 8])
        if rng.random() < 0.5:
            rep = rng.choice(["2 2 1", "2 2 2", "3 2 1", "1 2 2", "3 3 1"]); T = rng.choice([300, 900, 1500, 4000]); skin = rng.choice([0.4, 0.8, 2.0])
            steps = rng.choice([60, 120, 200]); sd = rng.randrange(1, 10**7)
            text = reb.replace("create_atoms 2 box basis 1 1 basis 2 1 basis 3 2 basis 4 2 basis 5 2 basis 6 2",
                               "create_atoms 2 box basis 1 1 basis 2 1 basis 3 2 basis 4 2 basis 5 2 basis 6 2\nreplicate " + rep)
            text = text.replace("thermo 10", f"velocity all create {T}.0 {sd}\nneighbor {skin} bin\nthermo {steps // 4}").replace("run 20", f"run {steps}")
            desc = f"rebomos rep {rep} T {T} skin {skin} steps {steps} seed {sd}"
        else:
            n = rng.choice([8, 10, 12, 14]); T = rng.choice([300, 863, 1400, 3000]); steps = rng.choice([60, 100, 160]); sd = rng.randrange(1, 10**7)
            frac = rng.choice([0.0075, 0.03, 0.08])
            text = aea.replace("region MeSi block 0 20 0 20 0 20", f"region MeSi block 0 {n} 0 {n} 0 {n}").replace("type/fraction 2 0.0075 7683797", f"type/fraction 2 {frac} {sd}")
            text = text.replace("velocity all create 863.0 1082337", f"velocity all create {T}.0 {sd}").replace("thermo 100", f"thermo {steps // 4}").replace("run 400", f"run {steps}")
            desc = f"aeam cells {n} frac {frac} T {T} steps {steps} seed {sd}"
        assert "fix integrate all nve" in text
        rc0, out0, err0 = _run(text)
        if np_ == 1:   # one rank: the fix in its default mode (the host reneighbors) and with one brick of the library's
            rc1, out1, err1 = _run(text.replace("fix integrate all nve", "fix integrate all nve/mdp"), env=env)
            rc2, out2, err2 = _run(text.replace("fix integrate all nve", "fix integrate all nve/mdp bricks yes"), env=env)
        else:
            rc1, out1, err1 = _run(text, np=np_)
            rc2, out2, err2 = _run(text.replace("fix integrate all nve", "fix integrate all nve/mdp"), np=np_, env=env)
        r0, r1, r2 = _thermo_rows(out0), _thermo_rows(out1), _thermo_rows(out2)
        ok = rc0 == 0 and rc1 == 0 and rc2 == 0 and rows_equal(r1, r0) and rows_equal(r2, r0)
        bad += 0 if ok else 1
        ren = out2.split("reneighborings on the device")[0].split()[-1] if "reneighborings on the device" in out2 else "?"
        print(f"{'ok ' if ok else 'BAD'} case {k} np {np_} {desc} rows {len(r0)} host builds {out0.split('Neighbor list builds = ')[-1].split()[0]} device reneighborings {ren}"
              + ("" if ok else f" rc {rc0} {rc1} {rc2} {err1[-200:]} {err2[-200:]}"), flush=True)
    print(f"{ncase} cases, {bad} bad, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)
main()
