"""ctypes binding of oracle/liboracle.so (CPU restatement of the reference hot path).
TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")

D22 = (C.c_double * 2) * 2


class RebomosParams(C.Structure):
    _fields_ = [(n, D22) for n in ("rcmin", "rcmax", "rcmaxsq", "Q", "alpha", "A", "BIJc", "Beta")] + [
        ("b", (C.c_double * 2) * 7), ("bg", (C.c_double * 2) * 7), ("a", (C.c_double * 2) * 4)] + [
        (n, D22) for n in ("rcLJmin", "rcLJmax", "epsilon", "sigma", "lj1", "lj2", "lj3", "lj4")] + [
        ("cut3rebo", C.c_double)]


def product_rebomos_params(P: "RebomosParams"):
    """the product's parameter struct (host/capi.RebomosParams) filled from the oracle's identically laid out
    leading fields -- for tests that feed both sides the same numbers"""
    from lammps_plugins_amd.host import capi
    out = capi.RebomosParams()
    for name, _ in capi.RebomosParams._fields_:
        setattr(out, name, getattr(P, name))
    return out


MAXEL = 16


class AeamPot(C.Structure):
    _fields_ = [
        ("nelements", C.c_int), ("nnonangular", C.c_int), ("nangular", C.c_int), ("nrhomax", C.c_int),
        ("nrmax", C.c_int),
        ("elements", (C.c_char * 16) * MAXEL),
        ("mass", C.c_double * MAXEL), ("drho", C.c_double * MAXEL),
        ("nrho", C.c_int * MAXEL),
        ("nr", (C.c_int * MAXEL) * MAXEL),
        ("dr", (C.c_double * MAXEL) * MAXEL), ("cut", (C.c_double * MAXEL) * MAXEL),
        ("nfrho", C.c_int), ("nrhor", C.c_int), ("nz2r", C.c_int),
        ("nrrho", C.c_int * (MAXEL * MAXEL)), ("nrz2r", C.c_int * (MAXEL * MAXEL)),
        ("drrho", C.c_double * (MAXEL * MAXEL)), ("drz2r", C.c_double * (MAXEL * MAXEL)),
        ("type2frho", C.c_int * (MAXEL + 1)),
        ("type2rhor", (C.c_int * (MAXEL + 1)) * (MAXEL + 1)),
        ("type2z2r", (C.c_int * (MAXEL + 1)) * (MAXEL + 1)),
        ("frho_spline", C.POINTER(C.c_double)), ("rhor_spline", C.POINTER(C.c_double)),
        ("z2r_spline", C.POINTER(C.c_double)),
    ]


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.rebomos_oracle_read_params.restype = C.c_int
        lib.rebomos_oracle_compute.restype = C.c_int
        lib.aeam_oracle_read.restype = C.c_int
        lib.aeam_oracle_compute.restype = C.c_int

    # ---------------- REBO-MoS ----------------
    def rebomos_params(self, filename) -> RebomosParams:
        P = RebomosParams()
        rc = self.lib.rebomos_oracle_read_params(filename.encode(), C.byref(P))
        if rc:
            raise RuntimeError(f"rebomos_oracle_read_params({filename}) -> {rc}")
        return P

    def rebomos_compute(self, P, nlocal, x, elem, tag, numneigh, offset, neigh, eflag=3, vflag=5, phases=3):
        nall = len(x)
        x = np.ascontiguousarray(x, dtype=np.float64)
        elem = np.ascontiguousarray(elem, dtype=np.int32)
        tag = np.ascontiguousarray(tag, dtype=np.int32)
        numneigh = np.ascontiguousarray(numneigh, dtype=np.int32)
        offset = np.ascontiguousarray(offset, dtype=np.int64)
        neigh = np.ascontiguousarray(neigh, dtype=np.int32)
        out = dict(f=np.zeros((nall, 3)), eng=C.c_double(0.0), virial_fdotr=np.zeros(6), virial_tally=np.zeros(6),
                   eatom=np.zeros(nall), vatom=np.zeros((nall, 6)), nM=np.zeros(nall), nS=np.zeros(nall),
                   rebo_numneigh=np.zeros(nall, dtype=np.int32))
        rc = self.lib.rebomos_oracle_compute(
            C.byref(P), C.c_int(nlocal), C.c_int(nall - nlocal), _p(x, C.c_double), _p(elem, C.c_int),
            _p(tag, C.c_int), _p(numneigh, C.c_int), _p(offset, C.c_longlong), _p(neigh, C.c_int),
            C.c_int(eflag), C.c_int(vflag), _p(out["f"], C.c_double), C.byref(out["eng"]),
            _p(out["virial_fdotr"], C.c_double), _p(out["virial_tally"], C.c_double), _p(out["eatom"], C.c_double),
            _p(out["vatom"], C.c_double), _p(out["nM"], C.c_double), _p(out["nS"], C.c_double),
            _p(out["rebo_numneigh"], C.c_int), C.c_int(phases))
        if rc:
            raise RuntimeError(f"rebomos_oracle_compute -> {rc}")
        out["eng"] = out["eng"].value
        return out

    # ---------------- AEAM ----------------
    def aeam_pot(self, filename) -> AeamPot:
        T = AeamPot()
        rc = self.lib.aeam_oracle_read(filename.encode(), C.byref(T))
        if rc:
            raise RuntimeError(f"aeam_oracle_read({filename}) -> {rc}")
        return T

    @staticmethod
    def aeam_splines(T: AeamPot):
        """numpy views of the oracle's spline tables: (frho[nfrho][nrhomax+1][7], rhor[...], z2r[...])"""
        fr = np.ctypeslib.as_array(T.frho_spline, shape=(T.nfrho, T.nrhomax + 1, 7))
        rh = np.ctypeslib.as_array(T.rhor_spline, shape=(T.nrhor, T.nrmax + 1, 7))
        z2 = np.ctypeslib.as_array(T.z2r_spline, shape=(T.nz2r, T.nrmax + 1, 7))
        return fr, rh, z2

    def aeam_compute(self, T, nlocal, x, type_, numneigh, offset, neigh, eflag=3, vflag=5):
        nall = len(x)
        x = np.ascontiguousarray(x, dtype=np.float64)
        type_ = np.ascontiguousarray(type_, dtype=np.int32)
        numneigh = np.ascontiguousarray(numneigh, dtype=np.int32)
        offset = np.ascontiguousarray(offset, dtype=np.int64)
        neigh = np.ascontiguousarray(neigh, dtype=np.int32)
        out = dict(f=np.zeros((nall, 3)), eng=C.c_double(0.0), virial_fdotr=np.zeros(6), virial_tally=np.zeros(6),
                   eatom=np.zeros(nall), vatom=np.zeros((nall, 6)), rho=np.zeros(nall), fp=np.zeros(nall))
        rc = self.lib.aeam_oracle_compute(
            C.byref(T), C.c_int(nlocal), C.c_int(nall - nlocal), _p(x, C.c_double), _p(type_, C.c_int),
            _p(numneigh, C.c_int), _p(offset, C.c_longlong), _p(neigh, C.c_int), C.c_int(eflag), C.c_int(vflag),
            _p(out["f"], C.c_double), C.byref(out["eng"]), _p(out["virial_fdotr"], C.c_double),
            _p(out["virial_tally"], C.c_double), _p(out["eatom"], C.c_double), _p(out["vatom"], C.c_double),
            _p(out["rho"], C.c_double), _p(out["fp"], C.c_double))
        if rc:
            raise RuntimeError(f"aeam_oracle_compute -> {rc}")
        out["eng"] = out["eng"].value
        return out


_cached = None


def load() -> Oracle:
    global _cached
    if _cached is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
        if (not os.path.exists(LIB)) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
            subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)
        _cached = Oracle(C.CDLL(LIB))
    return _cached


def fold_ghost_forces(f_all: np.ndarray, owner: np.ndarray, nlocal: int) -> np.ndarray:
    """LAMMPS reverse_comm of f for a single periodic domain: add each ghost's force to its owner."""
    f = f_all[:nlocal].copy()
    np.add.at(f, owner, f_all[nlocal:])
    return f
