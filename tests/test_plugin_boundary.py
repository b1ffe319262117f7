"""Drop-in boundary: the plugin .so files export exactly the one C symbol LAMMPS'
`plugin load` looks up, register the `pair` styles `rebomos` / `aeam`, keep the reference's
argument checks and messages, and refuse to run without a HIP device (no CPU fallback).
CPU part: no compute.  GPU part (marked): `plugin load` + `run` through the mini-host reproduce
USER-REBOMOS/log.rebomos-bulk.1:54-56 in host mode (x up / f down every step)."""
import json
import os
import re
import subprocess

import pytest

from conftest import GOLDEN, ROOT

PKG = os.path.join(ROOT, "lammps-plugins_amd")
MINILMP = os.path.join(PKG, "minilmp")


def _run(script_text=None, script_file=None, timeout=300, env=None, np=1):
    args = [MINILMP] + (["-np", str(np)] if np > 1 else []) + (["-in", script_file] if script_file else [])
    p = subprocess.run(args, input=script_text, capture_output=True, text=True, cwd=PKG, timeout=timeout,
                       env=dict(os.environ, **env) if env else None)
    return p.returncode, p.stdout, p.stderr


@pytest.mark.parametrize("so", ["rebomosplugin.so", "aeamplugin.so"])
def test_plugin_exports_only_lammpsplugin_init(so):
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(PKG, so)], capture_output=True, text=True).stdout
    c_syms = [l.split()[-1] for l in out.splitlines() if " T " in l and not l.split()[-1].startswith("_Z")
              and l.split()[-1] not in ("_init", "_fini")]
    assert c_syms == ["lammpsplugin_init"]
    # the style class is there with the reference's name
    cls = "PairREBOMoS" if so.startswith("rebomos") else "PairAEAM"
    assert re.search(r"_ZN9LAMMPS_NS%d%s7computeEii" % (len(cls), cls), out)


HEAD = """units metal
lattice fcc 4.045
region b block 0 2 0 2 0 2
create_box 2 b
create_atoms 1 box
mass 1 27.0
mass 2 28.0
"""


@pytest.mark.parametrize("script,msg", [
    ("plugin load rebomosplugin.so\n" + HEAD + "pair_style rebomos 1.0\n", "Illegal pair_style command"),
    ("plugin load rebomosplugin.so\n" + HEAD + "pair_style rebomos\npair_coeff * * ../tests/golden/potentials/MoS.REBO.set5b Mo\n",
     "Incorrect args for pair coefficients"),
    ("plugin load rebomosplugin.so\n" + HEAD + "pair_style rebomos\npair_coeff 1 1 ../tests/golden/potentials/MoS.REBO.set5b Mo S\n",
     "Incorrect args for pair coefficients"),
    ("plugin load rebomosplugin.so\n" + HEAD + "pair_style rebomos\npair_coeff * * ../tests/golden/potentials/MoS.REBO.set5b Mo W\n",
     "Incorrect args for pair coefficients"),
    ("plugin load rebomosplugin.so\n" + HEAD + "pair_style rebomos\npair_coeff * * nosuchfile Mo S\n", "potential file"),
    ("plugin load aeamplugin.so\n" + HEAD + "pair_style aeam x\n", "Illegal pair_style command"),
    ("plugin load aeamplugin.so\n" + HEAD + "pair_style aeam\npair_coeff * * ../tests/golden/potentials/AlSi.aeam Si Al\n",
     "no matching atom order of input file and potential file"),
    ("plugin load aeamplugin.so\n" + HEAD + "pair_style aeam\npair_coeff * * ../tests/golden/potentials/AlSi.aeam Al Cu\n",
     "No matching element in AEAM potential file"),
    ("plugin load aeamplugin.so\n" + HEAD + "pair_style aeam\npair_coeff * * nosuchfile Al Si\n", "Cannot open AEAM potential file"),
    (HEAD + "pair_style rebomos\n", "Unrecognized pair style"),
])
def test_argument_checks_keep_reference_messages(script, msg):
    rc, out, err = _run(script)
    assert rc == 1
    assert msg in err


def test_plugin_registers_and_refuses_to_run_without_gpu():
    from lammps_plugins_amd.host import capi
    rc, out, err = _run(script_file="examples/in.rebomos-bulk.mi355x")
    assert "Loaded 2 plugins from rebomosplugin.so" in out          # pair rebomos + fix nve/mdp
    assert "Created 288 atoms" in out
    if capi.lib().mdp_device_count() == 0:
        assert rc == 1 and "needs a HIP device; there is no CPU fallback" in err
    else:
        assert rc == 0


def _thermo_rows(out):
    rows = []
    grab = False
    for line in out.splitlines():
        if line.strip().startswith("Step"):
            grab = True
            continue
        if grab:
            parts = line.split()
            if not parts or not re.fullmatch(r"\d+", parts[0]):
                grab = False
                continue
            rows.append([float(p) for p in parts])
    return rows


@pytest.mark.gpu
def test_plugin_load_and_run_reproduce_reference_log():
    log = json.load(open(os.path.join(GOLDEN, "rebomos_bulk_log.json")))
    rc, out, err = _run(script_file="examples/in.rebomos-bulk.mi355x")
    assert rc == 0, err
    rows = _thermo_rows(out)
    assert [int(r[0]) for r in rows] == [0, 10, 20]
    for got, ref in zip(rows, log["thermo"]):
        # columns: step temp press pe ke cellgamma vol -- the 8 significant digits LAMMPS prints
        assert got[1] == pytest.approx(ref["temp"], abs=1e-5 * max(1.0, abs(ref["temp"])) * 1e-2 + 6e-6)
        assert got[2] == pytest.approx(ref["press"], abs=6e-3)
        assert got[3] == pytest.approx(ref["pe"], abs=6e-5)
        assert got[4] == pytest.approx(ref["ke"], abs=6e-8)
        assert got[5] == pytest.approx(113.40187, abs=1e-5)
        assert got[6] == pytest.approx(log["volume"], abs=1e-4)
    assert "FullNghs:  %d" % log["full_neighbors"] in out
    assert "Nghost:    %d" % log["nghost"] in out
    assert "Neighbor list builds = 0" in out


@pytest.mark.gpu
def test_aeam_plugin_runs_sample_system_and_conserves_energy():
    rc, out, err = _run(script_file="examples/in.aeam-alsi.mi355x", timeout=600)
    assert rc == 0, err
    rows = _thermo_rows(out)
    assert [int(r[0]) for r in rows] == [0, 100, 200, 300, 400]
    etot = [r[2] for r in rows]                      # step temp etotal pe vol press
    assert max(etot) - min(etot) < 32000 * 2e-5      # NVE drift/fluctuation per atom
    assert -3.45 < rows[0][3] / 32000 < -3.38        # PE/atom of the Al(Si) lattice at 0 K displacement
    assert rows[0][1] == pytest.approx(863.0, rel=1e-6)
