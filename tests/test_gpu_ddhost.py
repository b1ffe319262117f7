"""GPU: `ddhost` -- a C++ host that runs device-resident multi-GPU MD of the rebomos style through the C-ABI alone
(lammps-plugins_amd/minihost/ddhost.cpp: one brick per GPU, one thread per GPU, mdp_dd_comm_step_begin/_end; no Python in
the loop).  On this one-GPU box the ranks share the card through the RCCL test double (MDP_RCCL_LIBRARY).  Known answers:
log.rebomos-bulk.1:54-56 (one rank), log.rebomos-bulk.4:22,54-56,72-75 (2 x 2 x 1 ranks: same thermo rows, Nlocal 72 each,
Nghost 2768 / 2768 / 2775 / 2775)."""
import json
import os
import re
import struct
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN
from lammps_plugins_amd.host import capi, system as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "lammps-plugins_amd")


def _ddhost(args, double=False, timeout=600):
    exe = os.path.join(PKG, "ddhost")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", PKG, "ddhost"], check=True)
    env = dict(os.environ)
    if double:
        if not os.path.exists(capi.FAKE_RCCL):
            subprocess.run(["make", "-C", PKG, "rccl-double"], check=True)
        env.update(MDP_RCCL_LIBRARY=capi.FAKE_RCCL, MDP_FAKE_RCCL_TIMEOUT_S="60")
    p = subprocess.run([exe] + [str(a) for a in args], cwd=PKG, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    rows = [[float(w) for w in l.split()] for l in p.stdout.splitlines() if re.match(r"^\s*\d+\s+[-0-9.e+]+\s+[-0-9.e+]+", l)]
    return p.stdout, rows


def _dump(prefix, ranks):
    tags, x, v = [], [], []
    for r in range(ranks):
        raw = open(f"{prefix}.{r}", "rb").read()
        n = struct.unpack_from("i", raw, 0)[0]
        rec = np.frombuffer(raw, dtype=np.dtype([("tag", "<i4"), ("x", "<f8", 3), ("v", "<f8", 3)]), count=n, offset=4)
        tags.append(rec["tag"])
        x.append(rec["x"])
        v.append(rec["v"])
    tags = np.concatenate(tags)
    order = np.argsort(tags)
    assert np.array_equal(tags[order], np.arange(1, len(tags) + 1))      # every atom owned exactly once
    return np.concatenate(x)[order], np.concatenate(v)[order]


@pytest.mark.parametrize("ranks", [1, 4])
def test_ddhost_reproduces_the_reference_logs(ranks):
    log = json.load(open(os.path.join(GOLDEN, "rebomos_bulk_log.json")))
    out, rows = _ddhost(["-ranks", ranks, "-replicate", 1, 1, 1, "-steps", 20, "-thermo", 10], double=ranks > 1)
    assert [int(r[0]) for r in rows] == [0, 10, 20]
    for got, ref in zip(rows, log["thermo"]):
        assert got[1] == pytest.approx(ref["temp"], abs=6e-6)
        assert got[2] == pytest.approx(ref["press"], abs=6e-3)
        assert got[3] == pytest.approx(ref["pe"], abs=6e-5)
        assert got[4] == pytest.approx(ref["ke"], abs=6e-8)
    counts = [(int(a), int(b)) for a, b in re.findall(r"rank \d+: Nlocal (\d+)\s+Nghost (\d+)", out)]
    if ranks == 1:
        assert counts == [(288, 4285)]                                   # log.rebomos-bulk.1:72-74
    else:
        assert "2 by 2 by 1 processor grid" in out and "TEST DOUBLE" in out
        assert [c[0] for c in counts] == [72] * 4                        # log.rebomos-bulk.4:72
        assert sorted(c[1] for c in counts) == [2768, 2768, 2775, 2775]  # log.rebomos-bulk.4:73-75


@pytest.mark.parametrize("ranks", [2, 8])
def test_ddhost_hot_run_on_n_ranks_follows_its_one_rank_run(ranks, tmp_path):
    """3 x 3 x 2 replica at 300 K with a drift: the ranks reneighbor (and migrate atoms) by the flag in the halo, the overlap
    policy trial runs in the first steps; positions per atom equal the one-rank run's"""
    common = ["-replicate", 3, 3, 2, "-steps", 60, "-thermo", 20, "-temp", 300, "-drift", 60, -45, 30]
    o1, r1 = _ddhost(["-ranks", 1, "-dump", tmp_path / "one"] + common)
    on, rn = _ddhost(["-ranks", ranks, "-dump", tmp_path / "many"] + common, double=True)
    x1, v1 = _dump(tmp_path / "one", 1)
    xn, vn = _dump(tmp_path / "many", ranks)
    assert np.abs(vn - v1).max() < 1e-7
    for a, b in zip(rn, r1):
        assert a[3] == pytest.approx(b[3], rel=1e-10) and a[4] == pytest.approx(b[4], rel=1e-9)
        assert a[2] == pytest.approx(b[2], rel=1e-7, abs=1e-3)
    builds = int(re.search(r"Neighbor list builds = (\d+)", on).group(1))
    assert builds >= 3 and "Dangerous builds = 0" in on
    assert re.search(r"Overlap policy = (split|lead|blocking|first|inline)", on)
    box = S.replicate(S.rebomos_bulk_cell(), (3, 3, 2)).box
    dx = xn - x1
    dx -= np.round(box.x2lamda(dx + box.lo)) @ box.h.T                  # same atom, possibly another periodic image
    assert np.abs(dx).max() < 1e-8


@pytest.mark.parametrize("ranks", [2, 4])
def test_ddhost_aeam_on_n_ranks_follows_its_one_rank_run(ranks, tmp_path):
    """the aeam style in the same C++ host: 16^3 fcc cells (16 384 atoms, 3 % Si) at 863 K with a drift -- fp forward and
    ghost-force reverse exchanges behind the interior tiles, reneighborings by the flag in the halo"""
    common = ["-style", "aeam", "-replicate", 16, 16, 16, "-frac2", 0.03, "-steps", 48, "-thermo", 16, "-temp", 863,
              "-drift", 40, 25, -30]
    o1, r1 = _ddhost(["-ranks", 1, "-dump", tmp_path / "one"] + common)
    on, rn = _ddhost(["-ranks", ranks, "-dump", tmp_path / "many"] + common, double=True)
    x1, v1 = _dump(tmp_path / "one", 1)
    xn, vn = _dump(tmp_path / "many", ranks)
    assert np.abs(vn - v1).max() < 1e-7
    assert len(rn) == len(r1) == 4
    for a, b in zip(rn, r1):
        assert a[3] == pytest.approx(b[3], rel=1e-10) and a[4] == pytest.approx(b[4], rel=1e-9)
    box = S.fcc_cell(4.045, 16).box
    dx = xn - x1
    dx -= np.round(box.x2lamda(dx + box.lo)) @ box.h.T
    assert np.abs(dx).max() < 1e-8
    assert int(re.search(r"Neighbor list builds = (\d+)", on).group(1)) >= 3
