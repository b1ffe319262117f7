"""ctypes binding of libmdpair_hip.so (include/mdpair_hip.h).  Python never computes forces itself:
every call here lands in the HIP kernels, and loading fails loudly when the library is missing."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(PKG_DIR, "libmdpair_hip.so")

D22 = (C.c_double * 2) * 2


class RebomosParams(C.Structure):
    """mirror of mdp_rebomos_params"""
    _fields_ = [(n, D22) for n in ("rcmin", "rcmax", "rcmaxsq", "Q", "alpha", "A", "BIJc", "Beta")] + [
        ("b", (C.c_double * 2) * 7), ("bg", (C.c_double * 2) * 7), ("a", (C.c_double * 2) * 4)] + [
        (n, D22) for n in ("rcLJmin", "rcLJmax", "epsilon", "sigma", "lj1", "lj2", "lj3", "lj4")]


class AeamTables(C.Structure):
    """mirror of mdp_aeam_tables"""
    _fields_ = [("ntypes", C.c_int), ("nelements", C.c_int), ("nnonangular", C.c_int),
                ("nrhomax", C.c_int), ("nrmax", C.c_int), ("nfrho", C.c_int), ("nrhor", C.c_int), ("nz2r", C.c_int),
                ("nrho", C.POINTER(C.c_int)), ("drho", C.POINTER(C.c_double)),
                ("nr", C.POINTER(C.c_int)), ("dr", C.POINTER(C.c_double)), ("cut", C.POINTER(C.c_double)),
                ("type2frho", C.POINTER(C.c_int)), ("type2rhor", C.POINTER(C.c_int)), ("type2z2r", C.POINTER(C.c_int)),
                ("frho_spline", C.POINTER(C.c_double)), ("rhor_spline", C.POINTER(C.c_double)),
                ("z2r_spline", C.POINTER(C.c_double))]


class MdConfig(C.Structure):
    """mirror of mdp_md_config"""
    _fields_ = [("style", C.c_int), ("nlocal", C.c_int), ("nghost", C.c_int), ("ntypes", C.c_int),
                ("skin", C.c_double), ("dt", C.c_double), ("ftm2v", C.c_double), ("mvv2e", C.c_double),
                ("bbox_lo", C.c_double * 3), ("bbox_hi", C.c_double * 3), ("nghost_self", C.c_int),
                ("master_list", C.c_int)]


class DdConfig(C.Structure):
    """mirror of mdp_dd_config"""
    _fields_ = [("boxlo", C.c_double * 3), ("h", C.c_double * 6), ("procgrid", C.c_int * 3), ("rank", C.c_int),
                ("cutghost", C.c_double), ("self_remote", C.c_int), ("nonperiodic", C.c_int * 3)]


STYLE_REBOMOS, STYLE_AEAM = 1, 2
EXPORTS = [
    "mdp_abi_version", "mdp_device_count", "mdp_create", "mdp_destroy", "mdp_last_error", "mdp_set_stream",
    "mdp_sync", "mdp_rebomos_set_params", "mdp_rebomos_read_file", "mdp_rebomos_params_from_scalars",
    "mdp_aeam_set_tables", "mdp_aeam_file_read", "mdp_aeam_file_info", "mdp_aeam_file_build", "mdp_aeam_file_free", "mdp_set_atoms_host", "mdp_set_positions_host",
    "mdp_set_box_host", "mdp_host_ghosts_derived",
    "mdp_set_neighbors_host", "mdp_set_skin", "mdp_set_neighbors_csr_host", "mdp_rebomos_compute_host", "mdp_aeam_density_host",
    "mdp_aeam_force_host", "mdp_md_setup", "mdp_md_build_neighbors", "mdp_md_initial_integrate",
    "mdp_md_final_integrate", "mdp_md_final_initial_integrate", "mdp_md_compute", "mdp_md_compute_begin", "mdp_md_compute_end", "mdp_md_pack_x", "mdp_md_unpack_x", "mdp_md_pack_scalar",
    "mdp_md_unpack_scalar", "mdp_md_pack_ghost_f", "mdp_md_unpack_add_f", "mdp_md_fold_self_ghost_f",
    "mdp_md_aeam_density", "mdp_md_aeam_force", "mdp_md_thermo", "mdp_md_download", "mdp_md_upload_x", "mdp_md_ptr",
    "mdp_md_neighbor_stats", "mdp_md_class_stats", "mdp_md_prune_stats", "mdp_hnve_setup", "mdp_hnve_off", "mdp_hnve_upload_v",
    "mdp_hnve_initial", "mdp_hnve_final", "mdp_hnve_download", "mdp_rebomos_list_info", "mdp_set_timing", "mdp_get_timing",
    "mdp_device_bytes", "mdp_host_release", "mdp_rebomos_check_host_list",
    "mdp_dd_setup", "mdp_dd_reneighbor", "mdp_dd_migrate_begin", "mdp_dd_migrate_pack", "mdp_dd_migrate_end",
    "mdp_dd_borders_begin", "mdp_dd_borders_pack", "mdp_dd_borders_end", "mdp_dd_info", "mdp_dd_forward_pack",
    "mdp_dd_forward_unpack", "mdp_dd_forward_scalar_pack", "mdp_dd_forward_scalar_unpack", "mdp_dd_reverse_pack",
    "mdp_dd_reverse_unpack", "mdp_md_moved_async", "mdp_md_integrate_check", "mdp_md_download_int", "mdp_md_download_x_all",
    "mdp_dd_comm_unique_id", "mdp_dd_comm_library", "mdp_dd_comm_init", "mdp_dd_comm_destroy", "mdp_dd_comm_reneighbor",
    "mdp_dd_comm_forward_begin", "mdp_dd_comm_forward_end", "mdp_dd_comm_forward_scalar", "mdp_dd_comm_reverse",
    "mdp_dd_comm_allreduce", "mdp_dd_comm_step_begin", "mdp_dd_comm_step_end", "mdp_dd_comm_step_info", "mdp_aeam_device_lists", "mdp_rebomos_host_list", "mdp_aeam_check_host_list",
    "mdp_md_defer_final", "mdp_md_list_state", "mdp_md_aeam_force_begin", "mdp_md_aeam_state", "mdp_dd_comm_aeam_exchange_begin", "mdp_dd_comm_aeam_exchange_end",
]


class MdpError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"libmdpair_hip: error {code}: {text}")
        self.code = code


_lib = None


def lib():
    """load the C-ABI library; no fallback of any kind if it is missing"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} not built: run `python __graft_entry__.py` (make -C lammps-plugins_amd)")
        try:
            # ONE HIP runtime per process: torch ships its own libamdhip64 under the same soname.  Whichever copy is
            # loaded first serves both; loaded after ours, torch finds "No HIP GPUs".  So torch goes first.
            import torch  # noqa: F401
        except ImportError:
            pass
        # (MDP_LIB_PATH: another build of the same library -- A/B runs of two kernel versions on one box)
        _lib = C.CDLL(os.environ.get("MDP_LIB_PATH") or LIB_PATH)
        _lib.mdp_last_error.restype = C.c_char_p
        _lib.mdp_md_ptr.restype = C.c_void_p
        _lib.mdp_device_bytes.restype = C.c_double
    return _lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int))


def comm_library():
    """(name of the RCCL object the library binds, True when it is the test double of tests/native)"""
    buf = C.create_string_buffer(256)
    rc = lib().mdp_dd_comm_library(buf, C.c_int(256))
    if rc < 0:
        raise MdpError(rc, "no RCCL library could be loaded (MDP_RCCL_LIBRARY / librccl.so.1)")
    return buf.value.decode(), rc == 1


FAKE_RCCL = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "native",
                                          "libfake_rccl.so"))


def read_rebomos_file(path: str) -> RebomosParams:
    """product-side parser (csrc/potfile.cpp), mirrors PairREBOMoS::read_file"""
    p = RebomosParams()
    err = C.create_string_buffer(512)
    rc = lib().mdp_rebomos_read_file(path.encode(), C.byref(p), err, C.c_int(512))
    if rc:
        raise MdpError(rc, err.value.decode())
    return p


class AeamFile:
    """product-side AEAM potential file (csrc/potfile.cpp), mirrors PairAEAM::read_file + array2spline"""

    def __init__(self, path: str):
        self.h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = lib().mdp_aeam_file_read(path.encode(), C.byref(self.h), err, C.c_int(512))
        if rc:
            raise MdpError(rc, err.value.decode())
        ne, nn, na = C.c_int(), C.c_int(), C.c_int()
        mass = (C.c_double * 64)()
        names = C.create_string_buffer(64 * 16 + 16)
        lib().mdp_aeam_file_info(self.h, C.byref(ne), C.byref(nn), C.byref(na), mass, C.c_int(64), names,
                                 C.c_int(64 * 16 + 16))
        self.nelements, self.nnonangular, self.nangular = ne.value, nn.value, na.value
        self.mass = list(mass)[:ne.value]
        self.elements = names.value.decode().split()

    def build(self, ntypes: int | None = None, map_=None) -> AeamTables:
        ntypes = self.nelements if ntypes is None else ntypes
        m = np.ascontiguousarray([0] + list(range(ntypes)) if map_ is None else map_, dtype=np.int32)
        t = AeamTables()
        rc = lib().mdp_aeam_file_build(self.h, C.c_int(ntypes), _ip(m), C.byref(t))
        if rc:
            raise MdpError(rc, "mdp_aeam_file_build failed")
        return t

    def cut_table(self, t: AeamTables) -> np.ndarray:
        """(ntypes+1, ntypes+1) table of setfl->cut[i-1][j-1] (what init_one returns)"""
        ne = t.nelements
        cut = np.ctypeslib.as_array(t.cut, shape=(ne, ne))
        out = np.zeros((t.ntypes + 1, t.ntypes + 1))
        out[1:, 1:] = cut[:t.ntypes, :t.ntypes]
        return out

    def __del__(self):
        try:
            if self.h:
                lib().mdp_aeam_file_free(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass


class _SerialLib:
    """the library behind one process-wide lock: used when several ranks run as threads of one process
    (resident.run_ranks); RCCL runs have one process per GPU and call the library directly"""
    import threading as _threading
    lock = _threading.RLock()

    def __init__(self, L):
        self._L = L

    def __getattr__(self, name):
        fn = getattr(self._L, name)
        lock = self.lock

        def call(*a):
            with lock:
                return fn(*a)
        return call


class Context:
    """one mdp_ctx (one GPU sub-domain)"""
    serialize = False   # set by resident.run_ranks around threaded rehearsals

    def __init__(self, device: int = 0):
        self.L = _SerialLib(lib()) if Context.serialize else lib()
        self.h = C.c_void_p()
        rc = self.L.mdp_create(C.byref(self.h), C.c_int(device))
        if rc:
            raise MdpError(rc, "mdp_create failed (no usable HIP device?)")
        self._keep = []

    def close(self):
        if self.h:
            self.L.mdp_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc:
            raise MdpError(rc, self.L.mdp_last_error(self.h).decode())

    # ---------------- potentials
    def rebomos_set_params(self, params: RebomosParams):
        self._ck(self.L.mdp_rebomos_set_params(self.h, C.byref(params)))

    def aeam_set_tables(self, t: AeamTables):
        self._ck(self.L.mdp_aeam_set_tables(self.h, C.byref(t)))

    def set_stream(self, stream_ptr: int):
        self._ck(self.L.mdp_set_stream(self.h, C.c_void_p(stream_ptr)))

    def sync(self):
        self._ck(self.L.mdp_sync(self.h))

    def set_timing(self, on=True):
        self._ck(self.L.mdp_set_timing(self.h, C.c_int(1 if on else 0)))

    def get_timing(self):
        ms = (C.c_double * 8)()
        self._ck(self.L.mdp_get_timing(self.h, ms))
        return list(ms)

    # ---------------- host mode
    def set_atoms_host(self, nlocal, x_all, type_all, tag_all, ntypes, map_=None):
        x_all = np.ascontiguousarray(x_all, dtype=np.float64)
        type_all = np.ascontiguousarray(type_all, dtype=np.int32)
        tag_all = None if tag_all is None else np.ascontiguousarray(tag_all, dtype=np.int32)
        m = None if map_ is None else np.ascontiguousarray(map_, dtype=np.int32)
        self._ck(self.L.mdp_set_atoms_host(self.h, C.c_int(nlocal), C.c_int(len(x_all) - nlocal), _dp(x_all),
                                           _ip(type_all), _ip(tag_all), C.c_int(ntypes), _ip(m)))

    def set_positions_host(self, x_all):
        x_all = np.ascontiguousarray(x_all, dtype=np.float64)
        self._ck(self.L.mdp_set_positions_host(self.h, _dp(x_all)))

    def set_box_host(self, box):
        """Domain::h of the host's box (xprd, yprd, zprd, yz, xz, xy); None withdraws it"""
        if box is None:
            self._ck(self.L.mdp_set_box_host(self.h, None))
            return
        xy, xz, yz = box.tilt
        h = np.array([box.prd[0], box.prd[1], box.prd[2], yz, xz, xy], dtype=np.float64)
        self._ck(self.L.mdp_set_box_host(self.h, _dp(h)))

    def host_ghosts_derived(self):
        return bool(self.L.mdp_host_ghosts_derived(self.h))

    def set_neighbors_csr_host(self, numneigh, offset, neigh, skin):
        numneigh = np.ascontiguousarray(numneigh, dtype=np.int32)
        offset = np.ascontiguousarray(offset, dtype=np.int64)
        neigh = np.ascontiguousarray(neigh, dtype=np.int32)
        self._ck(self.L.mdp_set_neighbors_csr_host(self.h, C.c_int(len(numneigh)), _ip(numneigh),
                                                   offset.ctypes.data_as(C.POINTER(C.c_longlong)), _ip(neigh),
                                                   C.c_double(skin)))

    def set_skin(self, skin):
        self._ck(self.L.mdp_set_skin(self.h, C.c_double(skin)))

    def rebomos_host_list(self, on=True):
        self._ck(self.L.mdp_rebomos_host_list(self.h, C.c_int(1 if on else 0)))

    def aeam_device_lists(self, on=True):
        self._ck(self.L.mdp_aeam_device_lists(self.h, C.c_int(1 if on else 0)))

    def aeam_check_host_list(self, ilist, numneigh, rows, skin):
        ilist = np.ascontiguousarray(ilist, dtype=np.int32)
        numneigh = np.ascontiguousarray(numneigh, dtype=np.int32)
        ptrs = (C.POINTER(C.c_int) * len(rows))()
        for i, r in enumerate(rows):
            ptrs[i] = r.ctypes.data_as(C.POINTER(C.c_int))
        self._ck(self.L.mdp_aeam_check_host_list(self.h, C.c_int(len(ilist)), _ip(ilist), _ip(numneigh), ptrs,
                                                 C.c_double(skin)))

    def device_bytes(self) -> float:
        return float(self.L.mdp_device_bytes(self.h))

    def rebomos_check_host_list(self, ilist, numneigh, rows, cutneigh):
        """rows: list of int32 arrays, one per atom index (the LAMMPS firstneigh[] shape)"""
        ilist = np.ascontiguousarray(ilist, dtype=np.int32)
        numneigh = np.ascontiguousarray(numneigh, dtype=np.int32)
        ptrs = (C.POINTER(C.c_int) * len(rows))()
        for i, r in enumerate(rows):
            ptrs[i] = r.ctypes.data_as(C.POINTER(C.c_int))
        self._ck(self.L.mdp_rebomos_check_host_list(self.h, C.c_int(len(ilist)), _ip(ilist), _ip(numneigh), ptrs,
                                                    C.c_double(cutneigh)))

    def set_neighbors_paged_host(self, inum, gnum, ilist, numneigh, rows, skin):
        """rows: list of int32 arrays (one per atom index) -- exercises the LAMMPS int** path"""
        ilist = np.ascontiguousarray(ilist, dtype=np.int32)
        numneigh = np.ascontiguousarray(numneigh, dtype=np.int32)
        ptrs = (C.POINTER(C.c_int) * len(rows))()
        for i, r in enumerate(rows):
            ptrs[i] = r.ctypes.data_as(C.POINTER(C.c_int))
        self._ck(self.L.mdp_set_neighbors_host(self.h, C.c_int(inum), C.c_int(gnum), _ip(ilist), _ip(numneigh), ptrs,
                                               C.c_double(skin)))

    def rebomos_compute_host(self, nlocal, eflag=3, vflag=1):
        f = np.zeros((nlocal, 3))
        eng = C.c_double(0.0)
        vir = np.zeros(6)
        eatom = np.zeros(nlocal)
        vatom = np.zeros((nlocal, 6)) if vflag & 4 else None
        self._ck(self.L.mdp_rebomos_compute_host(self.h, C.c_int(eflag), C.c_int(vflag), _dp(f), C.byref(eng),
                                                 _dp(vir), _dp(eatom), _dp(vatom)))
        return dict(f=f, eng=eng.value, virial=vir, eatom=eatom, vatom=vatom)

    def aeam_density_host(self, nlocal, eflag=3, keep_fp=False):
        """keep_fp: fp (and rho) stay on the device (needs host_ghosts_derived())"""
        fp = None if keep_fp else np.zeros(nlocal)
        rho = None if keep_fp else np.zeros(nlocal)
        eng = C.c_double(0.0)
        eatom = np.zeros(nlocal)
        self._ck(self.L.mdp_aeam_density_host(self.h, C.c_int(eflag), _dp(fp), _dp(rho), C.byref(eng), _dp(eatom)))
        return dict(fp=fp, rho=rho, eng=eng.value, eatom=eatom)

    def aeam_force_host(self, nall, nlocal, fp_all, eflag=3, vflag=1):
        fp_all = None if fp_all is None else np.ascontiguousarray(fp_all, dtype=np.float64)
        f = np.zeros((nall, 3))
        eng = C.c_double(0.0)
        vir = np.zeros(6)
        eatom = np.zeros(nlocal)
        vatom = np.zeros((nall, 6)) if vflag & 4 else None
        self._ck(self.L.mdp_aeam_force_host(self.h, C.c_int(eflag), C.c_int(vflag), _dp(fp_all), _dp(f), C.byref(eng),
                                            _dp(vir), _dp(eatom), _dp(vatom)))
        return dict(f=f, eng=eng.value, virial=vir, eatom=eatom, vatom=vatom)

    # ---------------- resident mode
    def md_setup(self, cfg: MdConfig, x, v, type_, tag, mass, map_, ghost_owner, ghost_shift, ghost_type, ghost_tag):
        a = lambda arr, dt: np.ascontiguousarray(arr, dtype=dt)
        x, v, ghost_shift, mass = a(x, np.float64), a(v, np.float64), a(ghost_shift, np.float64), a(mass, np.float64)
        type_, tag, ghost_owner, ghost_type, ghost_tag = (a(type_, np.int32), a(tag, np.int32), a(ghost_owner, np.int32),
                                                          a(ghost_type, np.int32), a(ghost_tag, np.int32))
        m = None if map_ is None else a(map_, np.int32)
        self._ck(self.L.mdp_md_setup(self.h, C.byref(cfg), _dp(x), _dp(v), _ip(type_), _ip(tag), _dp(mass), _ip(m),
                                     _ip(ghost_owner), _dp(ghost_shift), _ip(ghost_type), _ip(ghost_tag)))

    def md_build_neighbors(self):
        self._ck(self.L.mdp_md_build_neighbors(self.h))

    def md_initial_integrate(self):
        self._ck(self.L.mdp_md_initial_integrate(self.h))

    def md_final_integrate(self):
        self._ck(self.L.mdp_md_final_integrate(self.h))

    def md_defer_final(self):
        self._ck(self.L.mdp_md_defer_final(self.h))

    def md_list_state(self):
        out = (C.c_double * 8)()
        self._ck(self.L.mdp_md_list_state(self.h, out))
        return dict(skin=out[0], inner_skin_cap=out[1], prune_buffer=out[2], late_builds=int(out[3]),
                    centre3_overflow=int(out[4]), centre3_list_mode=bool(out[5]))

    def md_final_initial_integrate(self):
        self._ck(self.L.mdp_md_final_initial_integrate(self.h))

    def md_compute(self, eflag=0, vflag=0):
        self._ck(self.L.mdp_md_compute(self.h, C.c_int(eflag), C.c_int(vflag)))

    def md_compute_begin(self, eflag=0, vflag=0):
        self._ck(self.L.mdp_md_compute_begin(self.h, C.c_int(eflag), C.c_int(vflag)))

    def md_compute_end(self, eflag=0, vflag=0):
        self._ck(self.L.mdp_md_compute_end(self.h, C.c_int(eflag), C.c_int(vflag)))

    def md_aeam_density(self, eflag=0):
        self._ck(self.L.mdp_md_aeam_density(self.h, C.c_int(eflag)))

    def md_aeam_force(self, eflag=0, vflag=0):
        self._ck(self.L.mdp_md_aeam_force(self.h, C.c_int(eflag), C.c_int(vflag)))

    def md_aeam_force_begin(self, eflag=0, vflag=0):
        self._ck(self.L.mdp_md_aeam_force_begin(self.h, C.c_int(eflag), C.c_int(vflag)))

    def md_aeam_state(self):
        """phases of the current aeam compute done so far, tiles that reach no remote ghost, tiles, whether ghost
        forces can be non-zero (see mdpair_hip.h)"""
        out = (C.c_int * 4)()
        self._ck(self.L.mdp_md_aeam_state(self.h, out))
        return dict(phase=int(out[0]), interior_tiles=int(out[1]), tiles=int(out[2]), ghost_forces=bool(out[3]))

    def md_thermo(self):
        out = (C.c_double * 9)()
        self._ck(self.L.mdp_md_thermo(self.h, out))
        o = list(out)
        return dict(ke=o[0], pe=o[1], virial=np.array(o[2:8]), maxdisp2=o[8])

    def md_download(self, nlocal, want=("x", "v", "f")):
        arrs = {k: (np.zeros((nlocal, 3)) if k in want else None) for k in ("x", "v", "f")}
        ea = np.zeros(nlocal) if "eatom" in want else None
        self._ck(self.L.mdp_md_download(self.h, _dp(arrs["x"]), _dp(arrs["v"]), _dp(arrs["f"]), _dp(ea)))
        arrs["eatom"] = ea
        return arrs

    def md_download_x_all(self, nall):
        x = np.zeros((max(nall, 1), 3))
        self._ck(self.L.mdp_md_download_x_all(self.h, _dp(x)))
        return x[:nall]

    def md_upload_x(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        self._ck(self.L.mdp_md_upload_x(self.h, _dp(x)))

    def md_ptr(self, name: str) -> int:
        p = self.L.mdp_md_ptr(self.h, name.encode())
        if not p:
            raise MdpError(-1, f"mdp_md_ptr({name}) returned NULL")
        return int(p)

    def rebomos_list_info(self):
        """shape of the style's Lennard-Jones lists after the last build (see mdpair_hip.h)"""
        out = (C.c_longlong * 8)()
        self._ck(self.L.mdp_rebomos_list_info(self.h, out))
        keys = ("tiled", "tiles", "union_stride", "union_max", "row_entries", "clusters", "large_tiles", "builds")
        return dict(zip(keys, (int(v) for v in out)))

    def md_neighbor_stats(self):
        out = (C.c_longlong * 8)()
        self._ck(self.L.mdp_md_neighbor_stats(self.h, out))
        return list(out)

    # ---------------- host mode with the integrator on the device (fix nve/mdp)
    def hnve_setup(self, dt, ftm2v, mass_per_type):
        m = np.ascontiguousarray(mass_per_type, dtype=np.float64)
        self._ck(self.L.mdp_hnve_setup(self.h, C.c_double(dt), C.c_double(ftm2v), _dp(m), C.c_int(len(m) - 1)))

    def hnve_off(self):
        self._ck(self.L.mdp_hnve_off(self.h))

    def hnve_upload_v(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        self._ck(self.L.mdp_hnve_upload_v(self.h, _dp(v)))

    def hnve_initial(self):
        m, d = C.c_int(0), C.c_int(0)
        self._ck(self.L.mdp_hnve_initial(self.h, C.byref(m), C.byref(d)))
        return bool(m.value), bool(d.value)

    def hnve_final(self):
        self._ck(self.L.mdp_hnve_final(self.h))

    def hnve_download(self, nlocal, want=("x", "v", "f")):
        out = {k: np.zeros((nlocal, 3)) for k in want}
        self._ck(self.L.mdp_hnve_download(self.h, _dp(out.get("x")), _dp(out.get("v")), _dp(out.get("f"))))
        return out

    def md_class_stats(self):
        """how the last compute's work was spread over the kernel classes (see mdpair_hip.h)"""
        out = (C.c_longlong * 32)()
        self._ck(self.L.mdp_md_class_stats(self.h, out))
        return [int(v) for v in out]

    def md_prune_stats(self):
        """dynamic pruning of the tile rows: prunings, late prunings, active, buffer (see mdpair_hip.h)"""
        out = (C.c_longlong * 4)()
        self._ck(self.L.mdp_md_prune_stats(self.h, out))
        return dict(prunings=int(out[0]), late=int(out[1]), active=bool(out[2]), buffer=out[3] * 1e-6)

    # ---------------- domain decomposition on the device (csrc/domain.hip)
    def dd_setup(self, box, procgrid, rank, cutghost, self_remote=False, nonperiodic=(0, 0, 0)):
        """box: host.system.Box (restricted triclinic); procgrid: bricks per dimension; nonperiodic[d] = 1: no
        periodic images / no wrap in dimension d (LAMMPS boundary f / s)"""
        cfg = DdConfig()
        xy, xz, yz = (float(t) for t in box.tilt)
        for d in range(3):
            cfg.boxlo[d] = float(box.lo[d])
            cfg.procgrid[d] = int(procgrid[d])
        for k, val in enumerate((box.prd[0], box.prd[1], box.prd[2], yz, xz, xy)):
            cfg.h[k] = float(val)
        cfg.rank, cfg.cutghost, cfg.self_remote = int(rank), float(cutghost), 1 if self_remote else 0
        for d in range(3):
            cfg.nonperiodic[d] = 1 if nonperiodic[d] else 0
        self._ck(self.L.mdp_dd_setup(self.h, C.byref(cfg)))
        self._dd_nranks = int(procgrid[0]) * int(procgrid[1]) * int(procgrid[2])

    def dd_reneighbor(self):
        self._ck(self.L.mdp_dd_reneighbor(self.h))

    def dd_migrate_begin(self):
        cnt = (C.c_int * self._dd_nranks)()
        self._ck(self.L.mdp_dd_migrate_begin(self.h, cnt))
        return np.array(cnt, dtype=np.int64)

    def dd_migrate_pack(self, d_buf):
        self._ck(self.L.mdp_dd_migrate_pack(self.h, C.c_void_p(d_buf)))

    def dd_migrate_end(self, narrive, d_buf):
        self._ck(self.L.mdp_dd_migrate_end(self.h, C.c_int(int(narrive)), C.c_void_p(d_buf)))

    def dd_borders_begin(self):
        cnt = (C.c_int * self._dd_nranks)()
        self._ck(self.L.mdp_dd_borders_begin(self.h, cnt))
        return np.array(cnt, dtype=np.int64)

    def dd_borders_pack(self, d_buf):
        self._ck(self.L.mdp_dd_borders_pack(self.h, C.c_void_p(d_buf)))

    def dd_borders_end(self, recv_counts, d_buf):
        rc = (C.c_int * self._dd_nranks)(*[int(v) for v in recv_counts])
        self._ck(self.L.mdp_dd_borders_end(self.h, rc, C.c_void_p(d_buf)))

    def dd_info(self):
        out = (C.c_longlong * 8)()
        sc, rc = (C.c_int * self._dd_nranks)(), (C.c_int * self._dd_nranks)()
        self._ck(self.L.mdp_dd_info(self.h, out, sc, rc))
        keys = ("nlocal", "nself", "nsend", "nrecv", "reneighbors", "left_last", "nranks", "rank")
        d = dict(zip(keys, (int(v) for v in out)))
        d["send_counts"], d["recv_counts"] = np.array(sc, dtype=np.int64), np.array(rc, dtype=np.int64)
        return d

    def dd_forward_pack(self, d_buf):
        self._ck(self.L.mdp_dd_forward_pack(self.h, C.c_void_p(d_buf)))

    def dd_forward_unpack(self, d_buf):
        self._ck(self.L.mdp_dd_forward_unpack(self.h, C.c_void_p(d_buf)))

    def dd_forward_scalar_pack(self, d_buf):
        self._ck(self.L.mdp_dd_forward_scalar_pack(self.h, C.c_void_p(d_buf)))

    def dd_forward_scalar_unpack(self, d_buf):
        self._ck(self.L.mdp_dd_forward_scalar_unpack(self.h, C.c_void_p(d_buf)))

    def dd_reverse_pack(self, d_buf):
        self._ck(self.L.mdp_dd_reverse_pack(self.h, C.c_void_p(d_buf)))

    def dd_reverse_unpack(self, d_buf):
        self._ck(self.L.mdp_dd_reverse_unpack(self.h, C.c_void_p(d_buf)))

    # ---------------- RCCL transport inside the library (csrc/comm_rccl.hip)
    def dd_comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        rc = self.L.mdp_dd_comm_unique_id(buf)
        if rc:
            raise MdpError(rc, "mdp_dd_comm_unique_id failed (RCCL not loadable?)")
        return buf.raw

    def dd_comm_init(self, id128: bytes):
        self._ck(self.L.mdp_dd_comm_init(self.h, C.c_char_p(id128)))

    def dd_comm_reneighbor(self):
        self._ck(self.L.mdp_dd_comm_reneighbor(self.h))

    def dd_comm_forward_begin(self):
        self._ck(self.L.mdp_dd_comm_forward_begin(self.h))

    def dd_comm_forward_end(self):
        self._ck(self.L.mdp_dd_comm_forward_end(self.h))

    def dd_comm_forward_scalar(self):
        self._ck(self.L.mdp_dd_comm_forward_scalar(self.h))

    def dd_comm_reverse(self):
        self._ck(self.L.mdp_dd_comm_reverse(self.h))

    def dd_comm_aeam_exchange_begin(self, with_reverse=True):
        self._ck(self.L.mdp_dd_comm_aeam_exchange_begin(self.h, C.c_int(1 if with_reverse else 0)))

    def dd_comm_aeam_exchange_end(self):
        self._ck(self.L.mdp_dd_comm_aeam_exchange_end(self.h))

    def dd_comm_step_begin(self, with_final=False, force_rebuild=-1, eflag=0, vflag=0):
        r = C.c_int(0)
        self._ck(self.L.mdp_dd_comm_step_begin(self.h, C.c_int(1 if with_final else 0), C.c_int(force_rebuild), C.c_int(eflag),
                                               C.c_int(vflag), C.byref(r)))
        return bool(r.value)

    def dd_comm_step_end(self, eflag=0, vflag=0, defer_final=False):
        self._ck(self.L.mdp_dd_comm_step_end(self.h, C.c_int(eflag), C.c_int(vflag), C.c_int(1 if defer_final else 0)))

    def dd_comm_step_info(self):
        out = (C.c_longlong * 8)()
        self._ck(self.L.mdp_dd_comm_step_info(self.h, out))
        names = {-1: "undecided (trial running)", 0: "split", 1: "lead", 2: "blocking", 3: "first", 4: "inline"}
        return dict(aeam_phased=int(out[0]), ghost_forces=bool(out[1]), reneighbored=bool(out[2]), reneighbors=int(out[3]),
                    dangerous=int(out[4]), overlap_policy=names.get(int(out[5]), str(int(out[5]))),
                    overlap_policy_fixed_by_env=bool(out[6]), overlap_policy_trial_ms=int(out[7]) * 1.0e-6)

    def dd_comm_allreduce(self, values, op=0):
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        self._ck(self.L.mdp_dd_comm_allreduce(self.h, _dp(v), C.c_int(len(v)), C.c_int(op)))
        return v

    def md_moved_async(self):
        """(moved, dangerous) of the check launched by the previous call; launches the next one"""
        m, d = C.c_int(0), C.c_int(0)
        self._ck(self.L.mdp_md_moved_async(self.h, C.byref(m), C.byref(d)))
        return bool(m.value), bool(d.value)

    def md_integrate_check(self, with_final=False):
        """initial_integrate (after the pending final half-kick if with_final) + md_moved_async in one kernel"""
        m, d = C.c_int(0), C.c_int(0)
        self._ck(self.L.mdp_md_integrate_check(self.h, C.c_int(1 if with_final else 0), C.byref(m), C.byref(d)))
        return bool(m.value), bool(d.value)

    def md_download_int(self, name: str, nlocal: int):
        out = np.zeros(max(nlocal, 1), dtype=np.int32)
        self._ck(self.L.mdp_md_download_int(self.h, name.encode(), _ip(out)))
        return out[:nlocal]

    # halo plumbing (device pointers as ints)
    def md_pack_x(self, n, d_sendlist, d_shift, d_buf):
        self._ck(self.L.mdp_md_pack_x(self.h, C.c_int(n), C.c_void_p(d_sendlist), C.c_void_p(d_shift), C.c_void_p(d_buf)))

    def md_unpack_x(self, first_ghost, n, d_buf):
        self._ck(self.L.mdp_md_unpack_x(self.h, C.c_int(first_ghost), C.c_int(n), C.c_void_p(d_buf)))

    def md_pack_scalar(self, which, n, d_sendlist, d_buf):
        self._ck(self.L.mdp_md_pack_scalar(self.h, C.c_int(which), C.c_int(n), C.c_void_p(d_sendlist), C.c_void_p(d_buf)))

    def md_unpack_scalar(self, which, first_ghost, n, d_buf):
        self._ck(self.L.mdp_md_unpack_scalar(self.h, C.c_int(which), C.c_int(first_ghost), C.c_int(n), C.c_void_p(d_buf)))

    def md_pack_ghost_f(self, first_ghost, n, d_buf):
        self._ck(self.L.mdp_md_pack_ghost_f(self.h, C.c_int(first_ghost), C.c_int(n), C.c_void_p(d_buf)))

    def md_unpack_add_f(self, n, d_sendlist, d_buf):
        self._ck(self.L.mdp_md_unpack_add_f(self.h, C.c_int(n), C.c_void_p(d_sendlist), C.c_void_p(d_buf)))

    def md_fold_self_ghost_f(self):
        self._ck(self.L.mdp_md_fold_self_ghost_f(self.h))
