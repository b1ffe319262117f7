"""The CPU pieces under AddressSanitizer + UBSan (`make -C lammps-plugins_amd asan-check`): potential-file front ends
on good and damaged files, the product's spline tables against the oracle's bit for bit, one oracle compute() per
style with every tally enabled on an isolated cluster, and the mini-host + plugin adapters up to the point where
they ask for a device.  Any sanitizer report or mismatch fails the target."""
import os
import subprocess

from conftest import ROOT


def test_cpu_pieces_are_clean_under_asan_and_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "lammps-plugins_amd"), "asan-check"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "asan_check: ok" in r.stdout
    assert r.stdout.count("mini-host under ASan") == 2
