"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
every symbol include/mdpair_hip.h declares.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from lammps_plugins_amd.host import capi


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mdpair_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mdp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/mdpair_hip.h but not exported"
    assert sorted(capi.EXPORTS) == declared
    assert L.mdp_abi_version() == 3


def test_struct_layouts_match_header():
    # 8 pair tables + b,bg (7x2) + a (4x2) + 8 pair tables, all doubles
    assert ctypes.sizeof(capi.RebomosParams) == 8 * (4 * 8 + 14 + 14 + 8 + 4 * 8)
    assert ctypes.sizeof(capi.MdConfig) == 4 * 4 + 8 * 4 + 8 * 6 + 8   # + nghost_self, master_list


def test_no_cpu_fallback_without_gpu():
    """on a box without a HIP device the product refuses to create a context (fails loudly)"""
    L = capi.lib()
    if L.mdp_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(capi.MdpError):
        capi.Context(0)
