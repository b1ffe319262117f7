#!/usr/bin/env python3
"""Device time of ONE step of ONE of the eight sub-domains of the 8-GPU headline run (REBO-MoS bulk, 24x24x24 replica
= 3,981,312 atoms, 2x2x2 bricks), measured on one GPU: all eight bricks are set up as threads of this process
(resident.ThreadTransport), then rank 0 alone steps -- pack, interior Lennard-Jones, unpack, centre kernels, boundary
tiles, gather, integrator -- against the ghost positions of the last exchange while the other ranks wait.  What the
timeline cannot show is the wire time of the all-to-all itself (5.1 MB per GPU and step), which the
real run overlaps with the interior tiles.
usage: rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 profiles/subdomain_step.py [nrep] [steps] [bricks] [style] [temp]
(bricks = 2, 4 or 8: the per-rank sub-domain of the 2-, 4- and 8-GPU runs; style = rebomos (default) or aeam: fcc cells
per dimension, 159 = the 16.1 M-atom alloy of BASELINE.json configs[4], started at 863 K as sample.in; its step has four
phases -- interior density | rest of the density, embedding, three-body forces | interior pair forces | the rest -- with
the packs and unpacks of the position, fp and ghost-force exchanges in between)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry

entry.load_package()
from lammps_plugins_amd.host import capi, resident, system as S

nrep = int(sys.argv[1]) if len(sys.argv) > 1 else 24
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
style = sys.argv[4] if len(sys.argv) > 4 else "rebomos"      # "aeam": nrep = fcc cells per dimension (159 = config #5)
temp = float(sys.argv[5]) if len(sys.argv) > 5 else (863.0 if style == "aeam" else 0.0)
POTS = os.path.join(ROOT, "tests", "golden", "potentials")
if style == "rebomos":
    s = S.replicate(S.rebomos_bulk_cell(), (nrep, nrep, nrep))
else:
    s = S.fcc_cell(4.045, (nrep, nrep, nrep), frac_type2=0.0075, seed=7683797)
    s.mass[1:3] = capi.AeamFile(os.path.join(POTS, "AlSi.aeam")).mass[:2]
v0 = S.gaussian_velocities(s, temp, seed=1082337) if temp > 0 else None
out = {}


def rank_fn(r, make_tr):
    import torch
    ctx = capi.Context(0)
    if style == "rebomos":
        p = capi.read_rebomos_file(os.path.join(POTS, "MoS.REBO.set5b"))
        ctx.rebomos_set_params(p)
        cutghost, skin, map_, st = 3.0 * p.rcmax[0][0] + 2.0, 2.0, [0, 0, 1], capi.STYLE_REBOMOS
    else:
        af = capi.AeamFile(os.path.join(POTS, "AlSi.aeam"))
        tabs = af.build()
        ctx.aeam_set_tables(tabs)
        ctx._af = (af, tabs)
        skin, map_, st = 1.0, None, capi.STYLE_AEAM
        cutghost = float(af.cut_table(tabs).max()) + skin
    tr = make_tr(ctx)
    d = resident.DeviceDomain(ctx, st, s, cutghost, skin, map_, v0=v0, transport=tr)
    d.compute(1, 1)
    th = d.thermo()
    for _ in range(3):
        d.step(0, 0)                      # real steps with exchanges: everything is warm
    # one full reneighboring of all eight bricks (collective): wall time on this shared GPU, an upper bound
    torch.cuda.synchronize()
    tr.sh.barrier.wait()
    tre = time.perf_counter()
    d.reneighbor()
    ctx.sync()
    tre = time.perf_counter() - tre
    d.compute(0, 0)
    for _ in range(2):
        d.step(0, 0)                      # real exchanges again: the halo buffers hold current ghost positions
    tr.sh.barrier.wait()
    if r == 0:
        out["reneighbor_wall_ms_bricks_sharing_one_gpu"] = round(tre * 1e3, 2)
        torch.cuda.synchronize()
        over0 = d.aeam_overlapped
        t0 = time.perf_counter()
        for k in range(steps):            # rank 0 alone; ghosts keep the positions of the last exchange
            if k == 0:
                ctx.md_initial_integrate()
            else:
                ctx.md_final_initial_integrate()   # (force-only steps: both half-kicks in one pass, as bench.py runs them)
            ctx.dd_forward_pack(d.send3.data_ptr())
            ctx.md_compute_begin(0, 0)
            ctx.dd_forward_unpack(d.recv3.data_ptr())
            if style == "rebomos":
                ctx.md_compute_end(0, 0)
                continue
            # aeam: the four phases of resident.DeviceDomain._aeam_step_compute without the wire
            ctx.md_aeam_density(0)
            if not ctx.md_aeam_state()["phase"] & 4:             # a row pruning was due: the blocking order
                ctx.dd_forward_scalar_pack(d.send1.data_ptr())
                ctx.dd_forward_scalar_unpack(d.recv1.data_ptr())
                ctx.md_aeam_force(0, 0)
                ctx.md_fold_self_ghost_f()
                ctx.dd_reverse_pack(d.rsend3.data_ptr())
                ctx.dd_reverse_unpack(d.rrecv3.data_ptr())
                continue
            d.aeam_overlapped += 1
            ctx.md_fold_self_ghost_f()
            ctx.dd_forward_scalar_pack(d.send1.data_ptr())
            if d.ghost_forces:
                ctx.dd_reverse_pack(d.rsend3.data_ptr())
            ctx.md_aeam_force_begin(0, 0)
            ctx.dd_forward_scalar_unpack(d.recv1.data_ptr())
            ctx.md_aeam_force(0, 0)
            if d.ghost_forces:
                ctx.dd_reverse_unpack(d.rrecv3.data_ptr())
        ctx.md_final_integrate()
        ctx.sync()
        dt = (time.perf_counter() - t0) / steps
        out.update(style=style, bricks=world, atoms_total=s.n, nlocal=d.nlocal, self_ghosts=d.nself, remote_ghosts=d.nrecv,
                   send_entries=d.nsend, halo_bytes_each_way=int(d.nsend * 24), pe_per_atom=th["pe"] / s.n, steps=steps,
                   ms_per_step_rank0_alone=round(dt * 1e3, 4))
        if style == "aeam":
            stt = ctx.md_aeam_state()
            out.update(tiles=stt["tiles"], interior_tiles=stt["interior_tiles"], ghost_forces=bool(d.ghost_forces),
                       steps_on_the_phased_path=d.aeam_overlapped - over0, prunings=ctx.md_prune_stats()["prunings"],
                       fp_bytes_each_way=int(d.nsend * 8), reverse_force_bytes=int(d.nrecv * 24) if d.ghost_forces else 0)
    tr.sh.barrier.wait()
    ctx.close()
    return None


resident.run_ranks(world, rank_fn)
print(json.dumps(out))
