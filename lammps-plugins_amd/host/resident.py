"""Resident-mode MD driver (host side): builds a sub-domain (owned atoms + ghost map), uploads it
once, then steps entirely on the GPU through the C-ABI.  This is the small slice of the LAMMPS host
around Pair::compute() that the bench and the tests need (Verlet::run loop of fix nve, thermo,
`neigh_modify every 1 check yes`); the arithmetic all happens in libmdpair_hip.so."""
from __future__ import annotations

import ctypes as C
import os as _os

import numpy as np

from . import capi
from . import system as S


def morton_order(x: np.ndarray, lo: np.ndarray, cell: float) -> np.ndarray:
    """argsort of atoms along a Z-order curve on a `cell`-sized grid (spatial locality for gathers)"""
    g = np.floor((x - lo) / cell).astype(np.int64)
    g -= g.min(axis=0)
    g = np.minimum(g, (1 << 20) - 1).astype(np.uint64)

    def spread(v):
        v = v & np.uint64(0x1FFFFF)
        v = (v | (v << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
        v = (v | (v << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
        v = (v | (v << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
        return v

    key = spread(g[:, 0]) | (spread(g[:, 1]) << np.uint64(1)) | (spread(g[:, 2]) << np.uint64(2))
    return np.argsort(key, kind="stable")


def hilbert_order(x: np.ndarray, lo: np.ndarray, cell: float) -> np.ndarray:
    """argsort of atoms along a 3-D Hilbert curve on a `cell`-sized grid.  Unlike the Z-order curve a
    Hilbert curve has no jumps: ANY run of consecutive atoms is a compact blob, which is what bounds the
    neighbour union of the Lennard-Jones tiles (csrc/rebomos.hip) and with it their LDS footprint.
    (Skilling's axes-to-transpose algorithm, vectorised.)"""
    if len(x) == 0:
        return np.zeros(0, dtype=np.int64)
    # exactly 2^bits cells per dimension over the atoms' extent (cells need not be cubic): the curve is
    # continuous only on its full cube, a partly occupied cube would bring the jumps back
    xmin = x.min(axis=0)
    ext = np.maximum(x.max(axis=0) - xmin, 1e-9)
    bits = max(1, int(np.ceil(np.log2(max(ext.max() / cell, 1.0)))))
    ncell = 1 << bits
    g = np.minimum(np.floor((x - xmin) / ext * ncell).astype(np.int64), ncell - 1)
    X = [g[:, 0].astype(np.uint64), g[:, 1].astype(np.uint64), g[:, 2].astype(np.uint64)]
    zero = np.uint64(0)
    q = 1 << (bits - 1)
    while q > 1:
        Q, P = np.uint64(q), np.uint64(q - 1)
        for i in range(3):
            hit = (X[i] & Q) != zero
            X[0] = np.where(hit, X[0] ^ P, X[0])
            t = np.where(hit, zero, (X[0] ^ X[i]) & P)
            X[0] = X[0] ^ t
            X[i] = X[i] ^ t
        q >>= 1
    X[1] ^= X[0]
    X[2] ^= X[1]
    t = np.zeros_like(X[0])
    q = 1 << (bits - 1)
    while q > 1:
        t = np.where((X[2] & np.uint64(q)) != zero, t ^ np.uint64(q - 1), t)
        q >>= 1
    X = [v ^ t for v in X]
    key = np.zeros_like(X[0])
    for b in range(bits - 1, -1, -1):
        for i in range(3):
            key = (key << np.uint64(1)) | ((X[i] >> np.uint64(b)) & np.uint64(1))
    return np.argsort(key, kind="stable")


def spatial_order(x: np.ndarray, lo: np.ndarray, cell: float, group=None, chunk: int = 96, box=None) -> np.ndarray:
    """The order atoms are stored in on the device: along a Hilbert curve (MDP_ORDER=morton: Z-order), and,
    when `group` (the atom types) is given, each stretch of `chunk` consecutive atoms additionally sorted by
    type.  The second step makes the 2-atom clusters and 32-atom tiles of the Lennard-Jones lists
    element-pure: pair cutoffs differ per element pair (Mo-Mo 10.5 A, S-S 7.8 A), a mixed cluster evaluates
    both atoms against the larger neighbourhood, and the four clusters sharing a wavefront all run as long as
    the longest list among them.  chunk = 3 tiles keeps every tile inside one compact stretch of the curve."""
    import os
    if box is not None:
        # Order in lamda (fractional) coordinates, rescaled to the edge lengths: in a TRICLINIC box (the
        # in.rebomos-bulk cell has an xy tilt of half an edge) the atoms fill a parallelepiped inside their
        # Cartesian bounding box and a curve over that box crosses its empty corners -- consecutive atoms jump.
        x = box.x2lamda(x) * np.linalg.norm(box.h, axis=0)
        lo = np.zeros(3)
    order = morton_order(x, lo, cell) if os.environ.get("MDP_ORDER", "hilbert") == "morton" else hilbert_order(x, lo, cell)
    if group is not None and chunk > 0 and len(order) and os.environ.get("MDP_ORDER_GROUP", "0") != "0":
        g = np.asarray(group)[order].astype(np.int64)
        key = (np.arange(len(order), dtype=np.int64) // chunk) * (int(g.max()) + 1) + g
        order = order[np.argsort(key, kind="stable")]
    return order


class Domain:
    """One GPU sub-domain in resident mode.

    owned atoms: x,v,type,tag (already restricted to this sub-domain)
    ghosts: owner (local index or -1), shift (Cartesian image shift, or absolute position if owner<0)
    """

    def __init__(self, ctx: capi.Context, style: int, box: S.Box, x, v, type_, tag, mass, map_, ghost_owner,
                 ghost_shift, ghost_type, ghost_tag, skin: float, dt: float = 0.001, master_list: bool = False):
        self.ctx, self.style, self.box = ctx, style, box
        self.nlocal, self.nghost = len(x), len(ghost_owner)
        self.mass = np.asarray(mass, dtype=np.float64)
        self.skin, self.dt = skin, dt
        xg = np.where((ghost_owner >= 0)[:, None], x[np.maximum(ghost_owner, 0)], 0.0) + ghost_shift \
            if self.nghost else np.zeros((0, 3))
        allx = np.concatenate([x, xg]) if self.nghost else x
        pad = 1.0 + skin
        cfg = capi.MdConfig()
        cfg.style, cfg.nlocal, cfg.nghost, cfg.ntypes = style, self.nlocal, self.nghost, len(mass) - 1
        cfg.skin, cfg.dt, cfg.ftm2v, cfg.mvv2e = skin, dt, S.FTM2V, S.MVV2E
        cfg.master_list = 1 if master_list else 0
        cfg.nghost_self = int((np.asarray(ghost_owner) >= 0).sum()) if self.nghost else 0
        lo, hi = allx.min(axis=0) - pad, allx.max(axis=0) + pad
        for d in range(3):
            cfg.bbox_lo[d], cfg.bbox_hi[d] = lo[d], hi[d]
        self.cfg = cfg
        ctx.md_setup(cfg, x, v, type_, tag, mass, map_, ghost_owner, ghost_shift, ghost_type, ghost_tag)
        self.natoms_total = self.nlocal  # overwritten by the multi-rank driver
        self.builds = 0

    @classmethod
    def single(cls, ctx, style, s: S.System, cutghost: float, skin: float, map_, v0=None, dt=0.001, sort=True,
               master_list=False):
        """whole periodic box on one GPU: ghosts are periodic self-images"""
        x = S.wrap(s.box, s.x)
        v = np.zeros_like(x) if v0 is None else np.asarray(v0, dtype=np.float64)
        t, g = s.type, s.tag
        if sort:
            order = spatial_order(x, s.box.lo, 3.0, group=s.type, box=s.box)
            x, v, t, g = x[order], v[order], t[order], g[order]
        owner, shift = S.make_ghosts(s.box, x, cutghost)
        shift_cart = S.mul_upper(shift, s.box.h)
        if sort and len(owner):
            go = spatial_order(x[owner] + shift_cart, s.box.lo - cutghost - 1.0, 3.0, box=s.box)
            owner, shift_cart = owner[go], shift_cart[go]
        d = cls(ctx, style, s.box, np.ascontiguousarray(x), np.ascontiguousarray(v), t, g, s.mass, map_,
                owner.astype(np.int32), np.ascontiguousarray(shift_cart), t[owner], g[owner], skin, dt,
                master_list=master_list)
        d.order_tag = g
        return d

    # ------------------------------------------------------------------ MD
    def build_neighbors(self):
        self.ctx.md_build_neighbors()
        self.builds += 1

    def compute(self, eflag=0, vflag=0):
        self.ctx.md_compute(eflag, vflag)

    def step(self, eflag=0, vflag=0):
        """one velocity-Verlet step (Verlet::run body): initial_integrate, [neighbor], force, final"""
        self.ctx.md_initial_integrate()
        self.ctx.md_compute(eflag, vflag)
        self.ctx.md_final_integrate()

    def thermo(self, natoms_total=None, volume=None):
        t = self.ctx.md_thermo()
        n = self.natoms_total if natoms_total is None else natoms_total
        vol = self.box.volume if volume is None else volume
        t["temp"] = S.temperature(t["ke"], n)
        t["press"] = S.pressure(t["ke"], t["virial"], n, vol)
        return t

    def needs_rebuild(self, thermo=None) -> bool:
        """`neigh_modify check yes`: any atom moved more than skin/2 since the last build"""
        t = self.ctx.md_thermo() if thermo is None else thermo
        return t["maxdisp2"] > (0.5 * self.skin) ** 2


class RankDomain(Domain):
    """one rank of a multi-GPU run: Domain + halo exchange over torch.distributed (RCCL on the GPU box)"""

    @classmethod
    def from_plan(cls, ctx, style, s: S.System, xw: np.ndarray, plan, skin, map_, v0=None, dt=0.001):
        """xw: wrapped positions of all atoms (same array the Decomposition was built from)"""
        own = plan.owned
        x = np.ascontiguousarray(xw[own])
        v = np.zeros_like(x) if v0 is None else np.ascontiguousarray(np.asarray(v0)[own])
        gshift = plan.ghost_shift.copy()
        remote = plan.ghost_owner_local < 0
        # remote ghosts carry their absolute start position in the shift slot (mdp_md_setup contract)
        gshift[remote] += xw[plan.ghost_global[remote]]
        d = cls(ctx, style, s.box, x, v, s.type[own], s.tag[own], s.mass, map_, plan.ghost_owner_local,
                np.ascontiguousarray(gshift), s.type[plan.ghost_global], s.tag[plan.ghost_global], skin, dt)
        d.plan = plan
        d.natoms_total = s.n
        d.halo = None
        return d

    def attach_halo(self, halo):
        self.halo = halo

    def forward_positions(self):
        h = self.halo
        if h is None or (h.nsend == 0 and h.nrecv == 0):
            return
        self.ctx.md_pack_x(h.nsend, h.sendlist.data_ptr(), h.sendshift.data_ptr(), h.send3.data_ptr())
        h.forward3()
        self.ctx.md_unpack_x(self.plan.nself, h.nrecv, h.recv3.data_ptr())

    def step_overlapped(self, eflag=0, vflag=0):
        """one step with the ghost-position exchange hidden behind the interior Lennard-Jones work
        (REBO-MoS): pack -> all_to_all (async) || compute_begin -> wait -> unpack -> compute_end"""
        h = self.halo
        self.ctx.md_initial_integrate()
        active = h is not None and (h.nsend or h.nrecv)
        work = None
        if active:
            self.ctx.md_pack_x(h.nsend, h.sendlist.data_ptr(), h.sendshift.data_ptr(), h.send3.data_ptr())
            work = h.forward3(async_op=True)
        self.ctx.md_compute_begin(eflag, vflag)
        if active:
            if work is not None:
                work.wait()
            self.ctx.md_unpack_x(self.plan.nself, h.nrecv, h.recv3.data_ptr())
        self.ctx.md_compute_end(eflag, vflag)
        self.ctx.md_final_integrate()

    def forward_fp(self):
        h = self.halo
        if h is None or (h.nsend == 0 and h.nrecv == 0):
            return
        self.ctx.md_pack_scalar(0, h.nsend, h.sendlist.data_ptr(), h.send1.data_ptr())
        h.forward1()
        self.ctx.md_unpack_scalar(0, self.plan.nself, h.nrecv, h.recv1.data_ptr())

    def reverse_forces(self):
        h = self.halo
        self.ctx.md_fold_self_ghost_f()
        if h is None or (h.nsend == 0 and h.nrecv == 0):
            return
        self.ctx.md_pack_ghost_f(self.plan.nself, h.nrecv, h.recv3.data_ptr())
        h.reverse3()
        self.ctx.md_unpack_add_f(h.nsend, h.sendlist.data_ptr(), h.send3.data_ptr())

    def compute(self, eflag=0, vflag=0):
        if self.style == capi.STYLE_REBOMOS:
            self.ctx.md_compute(eflag, vflag)
        else:
            self.ctx.md_aeam_density(eflag)
            self.forward_fp()
            self.ctx.md_aeam_force(eflag, vflag)
            self.reverse_forces()

    def step(self, eflag=0, vflag=0):
        if self.style == capi.STYLE_REBOMOS:
            return self.step_overlapped(eflag, vflag)
        self.ctx.md_initial_integrate()
        self.forward_positions()
        self.compute(eflag, vflag)
        self.ctx.md_final_integrate()


# ---------------------------------------------------------------------------------------------------
# reneighboring with re-derived ghosts (LAMMPS: Comm::exchange + Comm::borders at every rebuild).
# Rare (never in the 20-step reference run, log.rebomos-bulk.1:83) and therefore done the simple way:
# positions/velocities come back to the host, atoms are re-wrapped, re-assigned to bricks, ghosts
# re-derived and the sub-domain is uploaded again; the device then rebuilds and repacks its lists.
# ---------------------------------------------------------------------------------------------------

def gather_state(dom: Domain, s: S.System, dist=None, device=None):
    """global (x, v) in tag order from the resident state of all ranks"""
    got = dom.ctx.md_download(dom.nlocal, want=("x", "v"))
    tags = dom.tags_local
    x = np.zeros((s.n, 3))
    v = np.zeros((s.n, 3))
    if dist is None:
        x[tags - 1] = got["x"]
        v[tags - 1] = got["v"]
        return x, v
    import torch
    world = dist.get_world_size()
    nmax = torch.tensor([dom.nlocal], dtype=torch.int64, device="cpu" if getattr(dom, "stage_host", False) else device)
    dist.all_reduce(nmax, op=dist.ReduceOp.MAX)
    nmax = int(nmax.item())
    mine = torch.zeros((nmax, 7), dtype=torch.float64, device=device)
    mine[:dom.nlocal, 0] = torch.as_tensor(tags.astype(np.float64), device=device)
    mine[:dom.nlocal, 1:4] = torch.as_tensor(got["x"], device=device)
    mine[:dom.nlocal, 4:7] = torch.as_tensor(got["v"], device=device)
    if getattr(dom, "stage_host", False):
        mine = mine.cpu()
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    for p in parts:
        a = p.cpu().numpy()
        a = a[a[:, 0] > 0]
        idx = a[:, 0].astype(np.int64) - 1
        x[idx] = a[:, 1:4]
        v[idx] = a[:, 4:7]
    return x, v


def make_domain(ctx, style, s: S.System, cutghost, skin, map_, v0=None, dt=0.001, dist=None, device=None,
                stage_host=False):
    """(re)build the resident sub-domain of this rank from a global system"""
    if dist is None:
        d = Domain.single(ctx, style, s, cutghost, skin, map_, v0=v0, dt=dt)
        d.tags_local = d.order_tag
        d.natoms_total = s.n
        return d
    from . import decomp
    if not stage_host:
        # pack -> all_to_all -> unpack are only ordered when the context launches on the stream the collectives
        # synchronise with (torch's current stream)
        import torch
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    xw = S.wrap(s.box, s.x)
    dec = decomp.Decomposition(s.box, xw, dist.get_world_size(), cutghost, type_=s.type)
    plan = dec.plan(dist.get_rank())
    d = RankDomain.from_plan(ctx, style, s, xw, plan, skin, map_, v0=v0, dt=dt)
    d.tags_local = s.tag[plan.owned]
    d.attach_halo(decomp.Halo(plan, device, dist, stage_host=stage_host))
    d.stage_host = stage_host
    return d


def reneighbor(dom: Domain, s: S.System, cutghost, map_, dist=None, device=None) -> Domain:
    x, v = gather_state(dom, s, dist, device)
    s2 = S.System(s.box, x, s.type, s.tag, s.mass)
    d = make_domain(dom.ctx, dom.style, s2, cutghost, dom.skin, map_, v0=v, dt=dom.dt, dist=dist, device=device,
                    stage_host=getattr(dom, "stage_host", False))
    d.builds = dom.builds
    d.build_neighbors()
    return d


# ---------------------------------------------------------------------------------------------------
# Device-side domain decomposition (csrc/domain.hip): every rank keeps ONE brick, remaps / migrates /
# re-derives its ghosts on the GPU at each reneighboring and only exchanges counts and packed records.
# This is the path bench.py runs; Domain / RankDomain above (host-planned with numpy) stay as the
# independent reference the tests compare it with.
# ---------------------------------------------------------------------------------------------------

class Transport:
    """all-to-all of device buffers between the ranks of a torch.distributed group.  backend "nccl" is
    RCCL over xGMI (device buffers go straight in); any other backend is a rehearsal path for boxes with
    fewer GPUs than ranks: buffers are staged through the host around each collective."""

    def __init__(self, dist_module, device, stage_host: bool):
        import torch
        self.torch, self.dist, self.device, self.stage_host = torch, dist_module, device, stage_host
        self.on_gpu = torch.device(device).type == "cuda"
        self.world = dist_module.get_world_size()
        self.rank = dist_module.get_rank()

    def counts(self, send_counts: np.ndarray) -> np.ndarray:
        """recv[q] = what rank q sends to me.  Every rank gathers the whole count matrix (world^2 integers), so it
        also knows whether ANY rank has something to send: an exchange that is empty everywhere is skipped by all
        ranks alike (no collective is ever issued with empty tensors on every rank)."""
        torch = self.torch
        dev = "cpu" if self.stage_host else self.device
        s = torch.as_tensor(np.asarray(send_counts, dtype=np.int64), device=dev)
        rows = [torch.empty_like(s) for _ in range(self.world)]
        self.dist.all_gather(rows, s)
        m = torch.stack(rows).cpu().numpy()          # m[q][r]: rank q -> rank r
        self.last_total = int(m.sum())
        return m[:, self.rank].copy()

    def exchange(self, send, send_counts, recv_counts, width: int, recv=None, async_op=False):
        """send: flat float64 device tensor of sum(send_counts)*width; returns (recv tensor, work or None)"""
        torch = self.torch
        nrecv = int(np.sum(recv_counts)) * width
        if recv is None:
            recv = torch.empty(max(nrecv, 1), dtype=torch.float64, device=self.device)
        ins = [int(c) * width for c in send_counts]
        outs = [int(c) * width for c in recv_counts]
        nsend = sum(ins)
        if getattr(self, "last_total", 1) == 0 and nsend == 0 and nrecv == 0:
            return recv, None                    # nothing travels anywhere (decided from the gathered count matrix)
        if not self.stage_host:
            w = self.dist.all_to_all_single(recv[:nrecv], send[:nsend], outs, ins, async_op=async_op)
            return recv, w
        if self.on_gpu:
            torch.cuda.synchronize()
        o, i = recv[:nrecv].cpu(), send[:nsend].cpu()
        self.dist.all_to_all_single(o, i, outs, ins)
        recv[:nrecv].copy_(o)
        if self.on_gpu:
            torch.cuda.synchronize()
        return recv, None

    def any(self, flag: bool) -> bool:
        torch = self.torch
        t = torch.tensor([1.0 if flag else 0.0], device="cpu" if self.stage_host else self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return bool(t.item() > 0)

    def sum(self, values):
        torch = self.torch
        t = torch.tensor(list(values), dtype=torch.float64, device="cpu" if self.stage_host else self.device)
        self.dist.all_reduce(t)
        return t.cpu().numpy()


class NativeTransport:
    """The exchanges done INSIDE the library on RCCL (csrc/comm_rccl.hip: grouped ncclSend/ncclRecv between the
    bricks' GPUs) -- what a C++ host uses.  Python only bootstraps the communicator: rank 0 creates the 128-byte
    id, `bcast` hands it to every rank (torch.distributed broadcast, MPI_Bcast, ...)."""
    native = True
    stage_host = False

    def __init__(self, world: int, rank: int, bcast=None):
        self.world, self.rank = world, rank
        self.bcast = bcast if bcast is not None else (lambda b: b)
        self.ctx = None

    def attach(self, ctx: capi.Context):
        uid = ctx.dd_comm_unique_id() if self.rank == 0 else bytes(128)
        ctx.dd_comm_init(self.bcast(uid))
        self.ctx = ctx

    def any(self, flag: bool) -> bool:
        return bool(self.ctx.dd_comm_allreduce([1.0 if flag else 0.0], op=1)[0] > 0)

    def sum(self, values):
        return self.ctx.dd_comm_allreduce(list(values), op=0)


class DeviceDomain:
    """One brick of the periodic box per GPU, all bookkeeping on the device.

    s: the GLOBAL system (every rank builds the same synthetic system and keeps the atoms of its brick --
    setup only; nothing global exists afterwards).  transport=None: one GPU, no communication."""

    def __init__(self, ctx: capi.Context, style: int, s: S.System, cutghost: float, skin: float, map_, v0=None,
                 dt: float = 0.001, transport: Transport | None = None, master_list: bool = False,
                 self_remote: bool = False, nonperiodic=(0, 0, 0)):
        from . import decomp
        self.ctx, self.style, self.box, self.skin, self.dt = ctx, style, s.box, skin, dt
        self.tr = transport
        self.world = 1 if transport is None else transport.world
        self.rank = 0 if transport is None else transport.rank
        self.grid = decomp.proc_grid(self.world)
        self.natoms_total = s.n
        # (dimensions that are not periodic -- a slab, a free surface -- are neither wrapped nor given images)
        lam0 = s.box.x2lamda(s.x)
        per = np.array([0.0 if nonperiodic[d] else 1.0 for d in range(3)])
        xw = np.ascontiguousarray(s.box.lamda2x(lam0 - np.floor(lam0) * per))
        if self.world > 1:
            g = np.array(self.grid)
            lam = s.box.x2lamda(xw)
            cell = np.clip(np.floor(lam * g).astype(np.int64), 0, g - 1)
            mine = ((cell[:, 0] * g[1] + cell[:, 1]) * g[2] + cell[:, 2]) == self.rank
        else:
            mine = np.ones(s.n, dtype=bool)
        x = np.ascontiguousarray(xw[mine])
        v = np.zeros_like(x) if v0 is None else np.ascontiguousarray(np.asarray(v0, dtype=np.float64)[mine])
        cfg = capi.MdConfig()
        cfg.style, cfg.nlocal, cfg.nghost, cfg.ntypes = style, len(x), 0, len(s.mass) - 1
        cfg.skin, cfg.dt, cfg.ftm2v, cfg.mvv2e = skin, dt, S.FTM2V, S.MVV2E
        cfg.master_list = 1 if master_list else 0
        cfg.nghost_self = 0
        corners = s.box.lamda2x(np.array([[i, j, k] for i in (0, 1) for j in (0, 1) for k in (0, 1)], dtype=float))
        for d in range(3):   # provisional: the library sets the bounds of the brick at every reneighboring
            cfg.bbox_lo[d], cfg.bbox_hi[d] = corners[:, d].min() - cutghost - 2.0, corners[:, d].max() + cutghost + 2.0
        e3, e1 = np.zeros((0, 3)), np.zeros(0, dtype=np.int32)
        if transport is not None and not transport.stage_host and not getattr(transport, "native", False):
            # pack -> all_to_all -> unpack are only ordered when the context launches on the stream the
            # collectives synchronise with (torch's current stream)
            ctx.set_stream(transport.torch.cuda.current_stream().cuda_stream)
        ctx.md_setup(cfg, x, v, s.type[mine], s.tag[mine], s.mass, map_, e1, e3, e1, e1)
        ctx.dd_setup(s.box, self.grid, self.rank, cutghost, self_remote=self_remote, nonperiodic=nonperiodic)
        self.native = bool(getattr(transport, "native", False))
        if self.native:
            transport.attach(ctx)
        self.send3 = self.recv3 = self.send1 = self.recv1 = None
        self.builds = 0
        self.dangerous = 0
        self.reneighbor()

    # ------------------------------------------------------------------ reneighboring
    def reneighbor(self):
        """Comm::exchange + Comm::borders + Neighbor::build, per rank on the device"""
        ctx, tr = self.ctx, self.tr
        if tr is None:
            ctx.dd_reneighbor()
        elif self.native:
            ctx.dd_comm_reneighbor()
        else:
            torch = tr.torch
            f64 = dict(dtype=torch.float64, device=tr.device)
            sc = ctx.dd_migrate_begin()
            rc = tr.counts(sc)
            if _os.environ.get("MDP_DIAG"):
                print(f"[dd diag] rank {self.rank} reneighbor {self.builds}: nlocal {ctx.dd_info()['nlocal']} "
                      f"leave {sc.tolist()} arrive {rc.tolist()}", flush=True)
            send = torch.empty(max(int(sc.sum()) * 8, 1), **f64)
            ctx.dd_migrate_pack(send.data_ptr())
            recv, _ = tr.exchange(send, sc, rc, 8)
            if not tr.stage_host:
                torch.cuda.current_stream().synchronize()
            ctx.dd_migrate_end(int(rc.sum()), recv.data_ptr())
            sc = ctx.dd_borders_begin()
            rc = tr.counts(sc)
            send = torch.empty(max(int(sc.sum()) * 6, 1), **f64)
            ctx.dd_borders_pack(send.data_ptr())
            recv, _ = tr.exchange(send, sc, rc, 6)
            ctx.dd_borders_end(rc, recv.data_ptr())
            ctx.md_build_neighbors()
            self.send_counts, self.recv_counts = sc, rc
            ns, nr = int(sc.sum()), int(rc.sum())
            self.send3 = torch.empty(max(ns, 1) * 3, **f64)
            self.recv3 = torch.empty(max(nr, 1) * 3, **f64)
            self.send1 = torch.empty(max(ns, 1), **f64)
            self.recv1 = torch.empty(max(nr, 1), **f64)
            self._keep = (send, recv)   # until the stream has consumed them
        info = ctx.dd_info()
        self.nlocal, self.nself, self.nsend, self.nrecv = info["nlocal"], info["nself"], info["nsend"], info["nrecv"]
        self.nghost = self.nself + self.nrecv
        self.builds += 1
        self.fresh_ghosts = True

    @property
    def tags_local(self):
        return self.ctx.md_download_int("tag", self.nlocal)

    # ------------------------------------------------------------------ halo
    def _active(self):
        return self.tr is not None and (self.nsend or self.nrecv)

    def forward_positions(self, async_op=False):
        if self.native:
            self.ctx.dd_comm_forward_begin()      # exchange in flight on the library's communication stream
            if async_op:
                return "native"
            self.ctx.dd_comm_forward_end()
            return None
        self.ctx.dd_forward_pack(self.send3.data_ptr())
        _, w = self.tr.exchange(self.send3, self.send_counts, self.recv_counts, 3, recv=self.recv3, async_op=async_op)
        if async_op:
            return w
        self.ctx.dd_forward_unpack(self.recv3.data_ptr())

    def forward_fp(self):
        if self.native:
            self.ctx.dd_comm_forward_scalar()
            return
        self.ctx.dd_forward_scalar_pack(self.send1.data_ptr())
        self.tr.exchange(self.send1, self.send_counts, self.recv_counts, 1, recv=self.recv1)
        self.ctx.dd_forward_scalar_unpack(self.recv1.data_ptr())

    def reverse_forces(self):
        if self.native:
            self.ctx.dd_comm_reverse()            # folds the self-images too
            return
        self.ctx.md_fold_self_ghost_f()
        if not self._active():
            return
        self.ctx.dd_reverse_pack(self.recv3.data_ptr())
        self.tr.exchange(self.recv3, self.recv_counts, self.send_counts, 3, recv=self.send3)
        self.ctx.dd_reverse_unpack(self.send3.data_ptr())

    # ------------------------------------------------------------------ MD
    def compute(self, eflag=0, vflag=0):
        """Pair::compute for the current positions (ghosts must be current)"""
        if self.style == capi.STYLE_REBOMOS or not self._active():
            self.ctx.md_compute(eflag, vflag)
            return
        self.ctx.md_aeam_density(eflag)
        self.forward_fp()
        self.ctx.md_aeam_force(eflag, vflag)
        self.reverse_forces()

    def step(self, eflag=0, vflag=0, rebuild=False):
        """one velocity-Verlet step, Verlet::run order: initial_integrate, [reneighbor], forward comm, force,
        final_integrate.  REBO-MoS on several GPUs hides the ghost exchange behind the interior Lennard-Jones work."""
        ctx = self.ctx
        ctx.md_initial_integrate()
        if isinstance(rebuild, str):        # "auto": one GPU, deferred on-device flag read every step
            if self.tr is not None:
                raise ValueError("rebuild='auto' is for one-GPU runs; several ranks decide collectively (needs_rebuild)")
            rebuild = self.moved()
        if rebuild:
            self.reneighbor()               # the border exchange carries the current positions
        fresh, self.fresh_ghosts = self.fresh_ghosts and rebuild, False
        if not self._active():
            ctx.md_compute(eflag, vflag)
        elif self.style == capi.STYLE_REBOMOS:
            work = None if fresh else self.forward_positions(async_op=True)   # pack + all-to-all in flight
            ctx.md_compute_begin(eflag, vflag)
            if not fresh:
                if self.native:
                    ctx.dd_comm_forward_end()
                else:
                    if work is not None:
                        work.wait()
                    ctx.dd_forward_unpack(self.recv3.data_ptr())
            ctx.md_compute_end(eflag, vflag)
        else:
            if not fresh:
                self.forward_positions()
            self.compute(eflag, vflag)
        ctx.md_final_integrate()

    def thermo(self, reduce=True):
        """KE, PE, virial (summed over ranks), T and P of the whole system"""
        t = self.ctx.md_thermo()
        if self.tr is not None and reduce:
            tot = self.tr.sum([t["ke"], t["pe"], *t["virial"], ])
            t["ke"], t["pe"], t["virial"] = float(tot[0]), float(tot[1]), tot[2:8]
        t["temp"] = S.temperature(t["ke"], self.natoms_total)
        t["press"] = S.pressure(t["ke"], t["virial"], self.natoms_total, self.box.volume)
        return t

    def moved(self) -> bool:
        """deferred `neigh_modify check yes` flag (see mdp_md_moved_async); one GPU only -- several ranks must
        agree on the step they reneighbor at, see needs_rebuild"""
        m, d = self.ctx.md_moved_async()
        self.dangerous += int(d)
        return m

    def needs_rebuild(self, margin: float = 0.0) -> bool:
        """blocking check, collective over the ranks: some owned atom moved more than skin/2 - margin"""
        t = self.ctx.md_thermo()
        need = t["maxdisp2"] > max(0.5 * self.skin - margin, 0.25 * self.skin) ** 2
        return self.tr.any(need) if self.tr is not None else need


class ThreadTransport:
    """Rehearsal transport: N ranks as N threads of ONE process, each with its own context on the same GPU.
    A GPU box admits few processes on its card, so this is how the 4- and 8-brick decompositions (migration,
    borders, per-step halo) are exercised on one GPU; the rank code is exactly what runs over RCCL.
    Exchanges are device-to-device copies between the ranks' torch buffers, fenced by thread barriers."""

    class Shared:
        def __init__(self, world):
            import threading
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, shared: "ThreadTransport.Shared", rank: int, ctx: capi.Context, device):
        import torch
        self.torch, self.sh, self.rank, self.world, self.ctx, self.device = torch, shared, rank, shared.world, ctx, device
        self.stage_host = True   # every exchange is synchronous (no overlap in the rehearsal)

    def _all(self, value):
        self.sh.barrier.wait()
        self.sh.slots[self.rank] = value
        self.sh.barrier.wait()
        return list(self.sh.slots)

    def counts(self, send_counts):
        allc = self._all(np.asarray(send_counts, dtype=np.int64).copy())
        return np.array([allc[q][self.rank] for q in range(self.world)], dtype=np.int64)

    def exchange(self, send, send_counts, recv_counts, width, recv=None, async_op=False):
        torch = self.torch
        nrecv = int(np.sum(recv_counts)) * width
        if recv is None:
            recv = torch.empty(max(nrecv, 1), dtype=torch.float64, device=self.device)
        self.ctx.sync()                       # my pack kernel has finished
        off = np.concatenate([[0], np.cumsum(np.asarray(send_counts, dtype=np.int64) * width)])
        peers = self._all((send, off))
        at = 0
        for q in range(self.world):
            n = int(recv_counts[q]) * width
            if n:
                src, soff = peers[q]
                recv[at:at + n].copy_(src[int(soff[self.rank]):int(soff[self.rank]) + n])
            at += n
        torch.cuda.synchronize()
        self.sh.barrier.wait()                # nobody reuses a send buffer before every peer has copied
        return recv, None

    def any(self, flag):
        return any(self._all(bool(flag)))

    def sum(self, values):
        return np.sum(np.array(self._all(np.asarray(list(values), dtype=np.float64))), axis=0)


def run_ranks(world: int, fn, device=0):
    """run fn(rank, make_transport) on `world` threads; make_transport(ctx) gives the rank's ThreadTransport.
    Returns the list of results; the first exception of any rank is re-raised."""
    import threading
    import torch
    shared = ThreadTransport.Shared(world)
    out, err = [None] * world, [None] * world
    dev = torch.device("cuda", device)

    def body(r):
        try:
            torch.cuda.set_device(device)
            out[r] = fn(r, lambda ctx: ThreadTransport(shared, r, ctx, dev))
        except BaseException as e:   # noqa: BLE001 -- re-raised below
            err[r] = e
            shared.barrier.abort()

    import os
    old = capi.Context.serialize
    # every rank thread calls the library concurrently (own context, own stream): what a C++ host that drives
    # several GPUs from threads of one process does.  MDP_THREAD_SERIALIZE=1 puts one process-wide lock around the
    # library (debugging aid).
    capi.Context.serialize = os.environ.get("MDP_THREAD_SERIALIZE", "0") != "0"
    try:
        th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
    finally:
        capi.Context.serialize = old
    real = [e for e in err if e is not None and not isinstance(e, threading.BrokenBarrierError)]
    if real or any(e is not None for e in err):
        raise (real[0] if real else [e for e in err if e is not None][0])
    return out
