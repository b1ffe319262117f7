"""Writes the SURVEY Appendix-C fixture set under tests/golden/fixtures/ from THIS repository's CPU oracle.

    python tests/make_fixtures.py            (from the repository root; a few seconds, CPU only)

The numbers come from oracle/*.c -- not from the reference, which cannot be built here -- so the files pin nothing new
to the reference (see tests/fixture_cases.py); they freeze the oracle so that a co-edit of oracle and kernel fails
tests/test_golden_fixtures.py.  Re-run ONLY when the oracle is changed on purpose, and say why in the commit."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import conftest  # noqa: E402,F401  (registers the package and the paths)
import fixture_cases as FC  # noqa: E402
import oracle_bindings  # noqa: E402


def main():
    orc = oracle_bindings.load()
    P = orc.rebomos_params(conftest.POT_REBOMOS)
    T = orc.aeam_pot(conftest.POT_AEAM)
    os.makedirs(FC.FIXDIR, exist_ok=True)
    total = 0

    def write(name, style, s):
        nonlocal total
        eng = FC.engine(style, s, orc, P=P, T=T)
        out = FC.oracle_outputs(style, eng, s.x)
        path = os.path.join(FC.FIXDIR, name + ".npz")
        np.savez_compressed(path, **FC.pack(style, s, out))
        total += os.path.getsize(path)
        print(f"{name:20s} {style:8s} atoms {s.n:6d}  ghosts {eng.nghost:6d}  PE {float(out['eng']):.10f}  "
              f"{os.path.getsize(path) / 1024:.0f} KB")

    _, states = FC.bulk_states(orc, P)
    for step, s in states.items():
        write(f"R-bulk-0-step{step}", "rebomos", s)
    for name, (style, make) in FC.CASES.items():
        write(name, style, make())
    print(f"total {total / 1e6:.2f} MB in {FC.FIXDIR}")


if __name__ == "__main__":
    main()
