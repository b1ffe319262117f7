"""Host-planned sub-domains (TEST INFRASTRUCTURE): owned atoms, ghost map and halo lists planned with numpy
(lammps_plugins_amd.host.decomp), uploaded once, stepped through the same C-ABI.  This was the round-1 driver; it
stays as the INDEPENDENT reference the device-side domain decomposition (csrc/domain.hip, host/resident.py
DeviceDomain) is tested against -- other code, same answers."""
from __future__ import annotations

import numpy as np

from lammps_plugins_amd.host import capi, decomp
from lammps_plugins_amd.host import system as S
from lammps_plugins_amd.host.order import spatial_order


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
