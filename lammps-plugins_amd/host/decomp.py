"""Spatial domain decomposition + ghost-atom halo plan (the job LAMMPS' Comm brick does for the
reference: USER-REBOMOS/log.rebomos-bulk.4:22 "2 by 2 by 1 MPI processor grid").

One rank = one GPU = one brick in lamda (fractional) coordinates, so triclinic boxes work.
Every rank builds the same global plan from the same replicated input (deterministic, no
communication at setup); per step the only exchange is ONE all_to_all_single of ghost positions
(forward comm).  The device formulation is owner-computes, so REBO-MoS needs no reverse comm at
all; AEAM adds one scalar forward exchange (fp) and a reverse exchange of the ghost forces produced
by angular centres.  Transport: torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests)."""
from __future__ import annotations

import dataclasses

import numpy as np

from . import system as S


def proc_grid(nranks: int):
    """1 -> 1x1x1, 2 -> 2x1x1, 4 -> 2x2x1 (as log.rebomos-bulk.4:22), 8 -> 2x2x2"""
    table = {1: (1, 1, 1), 2: (2, 1, 1), 3: (3, 1, 1), 4: (2, 2, 1), 6: (3, 2, 1), 8: (2, 2, 2)}
    if nranks not in table:
        raise ValueError(f"unsupported rank count {nranks}")
    return table[nranks]


@dataclasses.dataclass
class RankPlan:
    rank: int
    owned: np.ndarray            # global indices of owned atoms, in local order
    ghost_owner_local: np.ndarray  # (nghost,) local index of owner if self-image else -1
    ghost_global: np.ndarray     # (nghost,) global atom index of each ghost
    ghost_shift: np.ndarray      # (nghost,3) Cartesian image shift
    nself: int                   # ghosts [0,nself) are self-images; remote ghosts follow grouped by source rank
    recv_counts: np.ndarray      # (nranks,) ghosts received from each rank (0 for self)
    send_counts: np.ndarray      # (nranks,)
    send_local: np.ndarray       # (sum send,) local indices to pack, grouped by destination rank
    send_shift: np.ndarray       # (sum send,3)


class Decomposition:
    def __init__(self, box: S.Box, x: np.ndarray, nranks: int, cutghost: float, sort_cell: float | None = 3.0,
                 type_=None):
        """x: wrapped positions of ALL atoms (identical on every rank); type_: their types (groups the
        storage order by element inside short stretches of the curve, see order.spatial_order)"""
        from .order import spatial_order
        self.box, self.nranks, self.cut = box, nranks, cutghost
        self.grid = np.array(proc_grid(nranks))
        lam = box.x2lamda(x)
        cell = np.minimum(np.floor(lam * self.grid).astype(np.int64), self.grid - 1)
        cell = np.maximum(cell, 0)
        self.rank_of = (cell[:, 0] * self.grid[1] + cell[:, 1]) * self.grid[2] + cell[:, 2]
        self.lam = lam
        natoms = len(x)
        self.local_index = np.zeros(natoms, dtype=np.int64)
        self.owned = []
        for r in range(nranks):
            idx = np.nonzero(self.rank_of == r)[0]
            if sort_cell and len(idx):
                idx = idx[spatial_order(x[idx], box.lo, sort_cell, group=None if type_ is None else np.asarray(type_)[idx], box=box)]
            self.owned.append(idx)
            self.local_index[idx] = np.arange(len(idx))
        # ghosts of every rank (every rank needs every other rank's list to know what to send)
        self.ghosts = []
        for r in range(nranks):
            sublo, subhi = self.sub_bounds(r)
            gidx, gshift = S.make_ghosts(box, x, cutghost, sublo, subhi, lam=lam, owned_mask=self.rank_of == r)
            src = self.rank_of[gidx]
            # order: self-images first, then by source rank; inside a group follow the source's local order
            key_rank = np.where(src == r, -1, src)
            order = np.lexsort((self.local_index[gidx], key_rank))
            self.ghosts.append((gidx[order], S.mul_upper(gshift[order], box.h), key_rank[order]))

    def sub_bounds(self, r):
        g = self.grid
        iz = r % g[2]
        iy = (r // g[2]) % g[1]
        ix = r // (g[1] * g[2])
        i = np.array([ix, iy, iz], dtype=float)
        return i / g, (i + 1) / g

    def plan(self, r: int) -> RankPlan:
        gidx, gshift, key_rank = self.ghosts[r]
        nself = int((key_rank == -1).sum())
        recv_counts = np.array([(key_rank == q).sum() for q in range(self.nranks)], dtype=np.int64)
        owner_local = np.where(key_rank == -1, self.local_index[gidx], -1).astype(np.int32)
        send_local, send_shift, send_counts = [], [], []
        for q in range(self.nranks):
            qidx, qshift, qkey = self.ghosts[q]
            m = qkey == r if q != r else np.zeros(len(qkey), bool)
            send_local.append(self.local_index[qidx[m]])
            send_shift.append(qshift[m])
            send_counts.append(int(m.sum()))
        return RankPlan(r, self.owned[r], owner_local, gidx, gshift, nself, recv_counts,
                        np.array(send_counts, dtype=np.int64),
                        np.concatenate(send_local).astype(np.int32) if send_local else np.zeros(0, np.int32),
                        np.concatenate(send_shift) if send_shift else np.zeros((0, 3)))


class Halo:
    """per-step exchanges on top of a RankPlan; buffers are torch tensors on `device`"""

    def __init__(self, plan: RankPlan, device, dist_module, stage_host: bool = False):
        """stage_host: move the buffers through host memory around each collective (lets a `gloo` group
        drive GPU-resident domains -- used by the 2-process single-GPU test; RCCL runs use device buffers)"""
        import torch
        self.torch, self.dist, self.plan = torch, dist_module, plan
        self.stage_host = stage_host
        self.nsend = int(plan.send_counts.sum())
        self.nrecv = int(plan.recv_counts.sum())
        f64 = dict(dtype=torch.float64, device=device)
        self.sendlist = torch.as_tensor(plan.send_local, dtype=torch.int32, device=device)
        self.sendshift = torch.as_tensor(np.ascontiguousarray(plan.send_shift), **f64).contiguous()
        self.send3 = torch.zeros(max(self.nsend, 1) * 3, **f64)
        self.recv3 = torch.zeros(max(self.nrecv, 1) * 3, **f64)
        self.send1 = torch.zeros(max(self.nsend, 1), **f64)
        self.recv1 = torch.zeros(max(self.nrecv, 1), **f64)
        self.in3 = [int(c) * 3 for c in plan.send_counts]
        self.out3 = [int(c) * 3 for c in plan.recv_counts]
        self.in1 = [int(c) for c in plan.send_counts]
        self.out1 = [int(c) for c in plan.recv_counts]

    def _a2a(self, out, inp, out_splits, in_splits, async_op=False):
        if not self.stage_host:
            return self.dist.all_to_all_single(out, inp, out_splits, in_splits, async_op=async_op)
        self.torch.cuda.synchronize()
        o, i = out.cpu(), inp.cpu()
        self.dist.all_to_all_single(o, i, out_splits, in_splits)
        out.copy_(o)
        self.torch.cuda.synchronize()

    def forward3(self, async_op=False):
        """send3 (packed owned positions+shift) -> recv3 (remote ghosts); async_op returns the work handle"""
        return self._a2a(self.recv3[:self.nrecv * 3], self.send3[:self.nsend * 3], self.out3, self.in3, async_op)

    def forward1(self):
        self._a2a(self.recv1[:self.nrecv], self.send1[:self.nsend], self.out1, self.in1)

    def reverse3(self):
        """recv3 (ghost forces, ghost order) -> send3 (contributions for my owned atoms, sendlist order)"""
        self._a2a(self.send3[:self.nsend * 3], self.recv3[:self.nrecv * 3], self.in3, self.out3)
