#!/usr/bin/env python3
"""Diagnosis of the concurrent-context failure of round 2 (gpurun_out/r2e3/pytest_domain_par.log): four rank
threads, one context each, NO process-wide lock (MDP_THREAD_SERIALIZE=0), library diagnostics on (MDP_DIAG=1).

  cold   the threads make the first GPU calls of the process (the situation of the failing log)
  warm   one single-context run first (every kernel of the path launched once), then the threads

One run each, in a fresh process: python3 profiles/r3_thread_diag.py cold|warm"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MDP_THREAD_SERIALIZE"] = "0"
os.environ["MDP_DIAG"] = "1"
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
import json  # noqa: E402

import test_gpu_domain as T  # noqa: E402
from lammps_plugins_amd.host import system as S  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "cold"
s = S.rebomos_bulk_cell()
if mode == "warm":
    T._run(1, s, None, 2, 0)
ok = True
for world in (4, 4, 8):
    try:
        r = T._run(world, s if world == 4 else S.replicate(s, (2, 2, 2)), None, 20, 5)
        print(f"[diag] {mode}: {world} threads ok, counts {r['counts0']}", flush=True)
    except Exception as e:  # noqa: BLE001
        ok = False
        print(f"[diag] {mode}: {world} threads FAILED: {e}", flush=True)
        traceback.print_exc()
        break
print(json.dumps({"mode": mode, "ok": ok}))
