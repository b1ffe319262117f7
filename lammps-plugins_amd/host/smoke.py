"""__graft_entry__.smoke(): one tiny REBO-MoS compute() on cuda:0 through the C-ABI, checked
against the CPU oracle (the oracle is the checker here, never the thing that produces the result)."""
from __future__ import annotations

import os
import sys

import numpy as np

from . import capi, system as S

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
POT = os.path.join(ROOT, "tests", "golden", "potentials", "MoS.REBO.set5b")


def run():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_bindings as ob
    import mdref
    orc = ob.load()
    P = orc.rebomos_params(POT)
    s = S.jitter(S.scale(S.rebomos_bulk_cell(), 1.05), 0.1, seed=7)
    eng = mdref.RebomosCPU(orc, P, s)
    ctx = capi.Context(0)                       # raises if there is no HIP device / library
    ctx.rebomos_set_params(capi.read_rebomos_file(POT))
    ctx.set_atoms_host(eng.nlocal, eng.all_positions(s.x), eng.type_all, eng.tag_all, 2, map_=[0, 0, 1])
    ctx.set_neighbors_csr_host(eng.nn, eng.off, eng.nb, 2.0)
    g = ctx.rebomos_compute_host(eng.nlocal, eflag=3, vflag=1)
    o = eng.compute(s.x)
    df = float(np.abs(g["f"] - o["f_owned"]).max())
    de = float(np.abs(g["eatom"] - o["eatom_owned"]).max())
    print(f"[smoke] REBO-MoS 288 atoms on cuda:0: PE={g['eng']:.6f} eV (oracle {o['eng']:.6f}), "
          f"max|dF|={df:.2e} eV/A, max|dE_atom|={de:.2e} eV")
    assert df < 1e-8 and de < 1e-8 and abs(g["eng"] - o["eng"]) < 1e-8 * abs(o["eng"])
    ctx.close()
