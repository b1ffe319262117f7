"""Resident-mode MD driver (host side): one brick of the box per GPU, set up once, then stepped entirely on the GPU
through the C-ABI.  This is the small slice of the LAMMPS host around Pair::compute() that the bench and the tests
need (Verlet::run loop of fix nve, thermo, `neigh_modify every 1 check yes`, the transport of the bricks' exchanges);
the arithmetic and the domain decomposition all happen in libmdpair_hip.so.  (The round-1 host-planned decomposition,
kept as the independent reference of the tests, lives in tests/hostplan.py.)"""
from __future__ import annotations

import ctypes as C
import os as _os

import numpy as np

from . import capi
from . import system as S



# ---------------------------------------------------------------------------------------------------
# Device-side domain decomposition (csrc/domain.hip): every rank keeps ONE brick, remaps / migrates /
# re-derives its ghosts on the GPU at each reneighboring and only exchanges counts and packed records.
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
        self.native_two_call = True    # (False: the library's transport through the piecewise calls below -- tests)
        if self.native:
            transport.attach(ctx)
        self.send3 = self.recv3 = self.send1 = self.recv1 = None
        self.builds = 0
        self.dangerous = 0
        self.aeam_overlapped = 0       # multi-GPU aeam steps whose exchanges travelled behind the interior tiles
        self._final_pending = False
        self.reneighbor()

    # ------------------------------------------------------------------ reneighboring
    def reneighbor(self):
        """Comm::exchange + Comm::borders + Neighbor::build, per rank on the device"""
        self.flush()
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
            if self.style == capi.STYLE_AEAM:   # the reverse exchange of a step travels beside the fp exchange
                self.rsend3 = torch.empty(max(nr, 1) * 3, **f64)
                self.rrecv3 = torch.empty(max(ns, 1) * 3, **f64)
            self._keep = (send, recv)   # until the stream has consumed them
        info = ctx.dd_info()
        self.nlocal, self.nself, self.nsend, self.nrecv = info["nlocal"], info["nself"], info["nsend"], info["nrecv"]
        self.nghost = self.nself + self.nrecv
        self.builds += 1
        self.fresh_ghosts = True
        # Whether a step exchanges anything is decided for ALL ranks alike, once per reneighboring: every exchange is a
        # collective over the whole group, so a rank without neighbours of its own (a brick facing vacuum across
        # non-periodic faces) takes part with empty messages as long as any rank has a halo.
        self.halo_active = tr.any(bool(self.nsend or self.nrecv)) if tr is not None else False
        if self.style == capi.STYLE_AEAM and tr is not None:
            # ghost forces are non-zero only when some rank has an angular centre next to a remote ghost: the reverse
            # exchange is skipped by all ranks alike otherwise (decided once per reneighboring)
            self.ghost_forces = tr.any(ctx.md_aeam_state()["ghost_forces"])

    @property
    def tags_local(self):
        return self.ctx.md_download_int("tag", self.nlocal)

    # ------------------------------------------------------------------ halo
    def _active(self):
        return self.tr is not None and self.halo_active

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

    def reverse_forces(self, remote=True):
        """ghost forces back to their owners.  remote=False: only the rank's own periodic images are folded -- the
        caller knows (collectively, `ghost_forces`) that no rank put a force on a remote ghost."""
        if self.native:
            if remote:
                self.ctx.dd_comm_reverse()        # folds the self-images too
            else:
                self.ctx.md_fold_self_ghost_f()
            return
        self.ctx.md_fold_self_ghost_f()
        if not remote or not self._active():
            return
        self.ctx.dd_reverse_pack(self.recv3.data_ptr())
        self.tr.exchange(self.recv3, self.recv_counts, self.send_counts, 3, recv=self.send3)
        self.ctx.dd_reverse_unpack(self.send3.data_ptr())

    # ------------------------------------------------------------------ MD
    def compute(self, eflag=0, vflag=0):
        """Pair::compute for the current positions (ghosts must be current)"""
        self.flush()
        if self.style == capi.STYLE_REBOMOS or not self._active():
            self.ctx.md_compute(eflag, vflag)
            return
        self.ctx.md_aeam_density(eflag)
        self.forward_fp()
        self.ctx.md_aeam_force(eflag, vflag)
        self.reverse_forces()

    def flush(self):
        """the final_integrate a step(defer_final=True) left to the next step's fused kernel, now (before anything
        that reads the velocities or replaces the forces)"""
        if self._final_pending:
            self.ctx.md_final_integrate()
            self._final_pending = False

    def _aeam_step_compute(self, eflag, vflag, fresh):
        """Pair::compute of a multi-GPU aeam step in four phases (mdpair_hip.h): the position exchange travels behind
        the density of the interior tiles, the style's fp exchange (pair_aeam.cpp:307) and the reverse exchange of the
        three-body forces on ghosts behind the pair forces of the interior tiles."""
        ctx, tr = self.ctx, self.tr
        work = None if fresh else self.forward_positions(async_op=True)
        ctx.md_compute_begin(eflag, vflag)                       # A: density, interior tiles
        if not fresh:
            if self.native:
                ctx.dd_comm_forward_end()
            else:
                if work not in (None, "native"):
                    work.wait()
                ctx.dd_forward_unpack(self.recv3.data_ptr())
        ctx.md_aeam_density(eflag)                               # B: the rest of passes 1 + 2 (+ three-body forces)
        rev = self.ghost_forces
        if not ctx.md_aeam_state()["phase"] & 4:                 # phase A did not run (pruning due, CSR lists, ...)
            # That is THIS rank's business (its own rows are due for pruning): the peers may be on the phased order in
            # the same step, so the exchanges issued here are exactly theirs, in their order -- fp forward, then the
            # ghost forces back iff `ghost_forces` says any rank has some (decided collectively per reneighboring).
            self.forward_fp()
            ctx.md_aeam_force(eflag, vflag)
            self.reverse_forces(remote=rev)
            return
        self.aeam_overlapped += 1
        if self.native:
            ctx.dd_comm_aeam_exchange_begin(rev)                 # fp out, ghost forces back: one group of sends
            ctx.md_aeam_force_begin(eflag, vflag)                # C: pair forces, interior tiles
            ctx.dd_comm_aeam_exchange_end()
            ctx.md_aeam_force(eflag, vflag)                      # D: the rest
            return
        ctx.md_fold_self_ghost_f()
        ctx.dd_forward_scalar_pack(self.send1.data_ptr())
        if rev:
            ctx.dd_reverse_pack(self.rsend3.data_ptr())
        _, w1 = tr.exchange(self.send1, self.send_counts, self.recv_counts, 1, recv=self.recv1, async_op=True)
        w3 = None
        if rev:
            _, w3 = tr.exchange(self.rsend3, self.recv_counts, self.send_counts, 3, recv=self.rrecv3, async_op=True)
        ctx.md_aeam_force_begin(eflag, vflag)                    # C
        if w1 is not None:
            w1.wait()
        ctx.dd_forward_scalar_unpack(self.recv1.data_ptr())
        ctx.md_aeam_force(eflag, vflag)                          # D
        if rev:
            if w3 is not None:
                w3.wait()
            ctx.dd_reverse_unpack(self.rrecv3.data_ptr())

    def step(self, eflag=0, vflag=0, rebuild=False, defer_final=False):
        """one velocity-Verlet step, Verlet::run order: initial_integrate, [reneighbor], forward comm, force,
        final_integrate.  REBO-MoS on several GPUs hides the ghost exchange behind the interior Lennard-Jones work.
        defer_final: leave this step's final_integrate to the next step, whose first kernel then does both
        half-kicks in one pass (mdp_md_final_initial_integrate; same arithmetic) -- for steps after which nothing
        reads the velocities; thermo() / compute() / reneighbor() / flush() complete it otherwise."""
        ctx = self.ctx
        if self.native and self.tr is not None and self.native_two_call:
            # The library's own transport drives the whole step in two calls (mdp_dd_comm_step_begin / _end): integrate,
            # decide, reneighbor or exchange, compute, final kick.  rebuild="halo" (or "auto"): the collective `check yes`
            # decision comes from the word that travelled with the previous step's halo -- no blocking call here.
            force = -1 if isinstance(rebuild, str) else (1 if rebuild else 0)
            ren = ctx.dd_comm_step_begin(self._final_pending, force, eflag, vflag)
            self._final_pending = False
            if ren:
                info = ctx.dd_info()
                self.nlocal, self.nself, self.nsend, self.nrecv = info["nlocal"], info["nself"], info["nsend"], info["nrecv"]
                self.nghost = self.nself + self.nrecv
                self.builds += 1
            ctx.dd_comm_step_end(eflag, vflag, defer_final)
            self._final_pending = bool(defer_final)
            si = ctx.dd_comm_step_info()
            self.aeam_overlapped, self.ghost_forces, self.dangerous = si["aeam_phased"], si["ghost_forces"], si["dangerous"]
            self.overlap_policy = si["overlap_policy"]
            return
        if isinstance(rebuild, str):        # "auto": one GPU, deferred on-device flag read every step
            if self.tr is not None:
                raise ValueError("rebuild='auto' is for one-GPU runs; the torch / thread transports decide collectively "
                                 "(needs_rebuild); the library's transport takes rebuild='halo'")
            rebuild, late = ctx.md_integrate_check(self._final_pending)   # integrate + check in one pass
            self.dangerous += int(late)
        elif self._final_pending:
            ctx.md_final_initial_integrate()
        else:
            ctx.md_initial_integrate()
        self._final_pending = False
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
            self._aeam_step_compute(eflag, vflag, fresh)
        if defer_final:
            self._final_pending = True
            ctx.md_defer_final()        # (the library completes the kick itself if velocities are read before the next step)
        else:
            ctx.md_final_integrate()

    def tune_overlap(self, max_steps=80):
        """library transport: force-only steps until the library's overlap-policy trial has chosen (comm_rccl.hip); returns
        the step info.  Collective: every rank runs the same steps."""
        si = self.ctx.dd_comm_step_info()
        n = 0
        while si["overlap_policy"].startswith("undecided") and n < max_steps:
            self.step(0, 0, rebuild="halo", defer_final=True)
            si = self.ctx.dd_comm_step_info()
            n += 1
        self.flush()
        si["trial_steps"] = n
        return si

    def thermo(self, reduce=True):
        """KE, PE, virial (summed over ranks), T and P of the whole system"""
        self.flush()
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
        self.flush()
        t = self.ctx.md_thermo()
        need = t["maxdisp2"] > max(0.5 * self.skin - margin, 0.25 * self.skin) ** 2
        if t["maxdisp2"] > (0.5 * self.skin) ** 2:
            self.dangerous += 1              # (an atom of THIS rank was beyond half the skin already: a late build)
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
            # (a timeout: ranks whose exchange schedules diverge fail the test with BrokenBarrierError instead of hanging)
            self.barrier = threading.Barrier(world, timeout=float(_os.environ.get("MDP_THREAD_BARRIER_TIMEOUT", "180")))
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


def run_ranks(world: int, fn, device=0, native=False):
    """run fn(rank, make_transport) on `world` threads; make_transport(ctx) gives the rank's ThreadTransport, or --
    native=True -- a NativeTransport: the library's own RCCL transport with `world` ranks.  On one GPU that needs
    MDP_RCCL_LIBRARY to name the test double of tests/native (RCCL itself refuses two ranks on one device).
    Returns the list of results; the first exception of any rank is re-raised."""
    import threading
    import torch
    shared = ThreadTransport.Shared(world)
    out, err = [None] * world, [None] * world
    dev = torch.device("cuda", device)

    def bcast_from(r):
        def bcast(b):                        # rank 0's communicator id to every rank thread
            shared.barrier.wait()
            if r == 0:
                shared.slots[0] = b
            shared.barrier.wait()
            return shared.slots[0]
        return bcast

    def body(r):
        try:
            torch.cuda.set_device(device)
            if native:
                out[r] = fn(r, lambda ctx: NativeTransport(world, r, bcast_from(r)))
            else:
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
