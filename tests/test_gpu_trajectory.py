"""GPU: long device-resident trajectories against the same trajectories driven on the host with ORACLE forces.

The reference's only dynamic known answer is 20 steps from 0 K (log.rebomos-bulk.1:54-56).  These tests pin the long-run
behaviour of the device path -- hundreds of steps from a thermal start, across the style's own list rebuilds, row
prunings, on-device reneighborings with re-sorting of the atoms -- to the CPU restatement of the reference: the host loop
below is fix nve around the oracle's compute() (tests/mdref.py), with its neighbour list rebuilt from scratch every
`rebuild_every` steps.  Positions must agree to 1e-8 A and the total energy to 1e-9 eV per atom at every sample, so any
energy drift of the device run is the potential's own (REBO-MoS: the Lennard-Jones term is not shifted at 2.5 sigma,
pair_rebomos.cpp:518-543), not the lists', the pruning's or the integrator's."""
import numpy as np
import pytest

from conftest import POT_AEAM, POT_REBOMOS
from lammps_plugins_amd.host import capi, resident, system as S
import mdref

pytestmark = pytest.mark.gpu


def _host_run(make_engine, s, v0, nsteps, every, skin, rebuild_every, dt=0.001):
    """velocity-Verlet around the oracle; returns {step: (x by tag, total energy, PE)}"""
    m = s.mass[s.type][:, None]
    x = S.wrap(s.box, s.x)
    v = v0.copy()
    eng = make_engine(S.System(s.box, x.copy(), s.type, s.tag, s.mass))
    x_built = x.copy()
    o = eng.compute(x, eflag=1, vflag=0)
    f = o["f_owned"]
    out = {0: (x.copy(), o["eng"] + S.kinetic_energy(m[:, 0], v), o["eng"])}
    for step in range(1, nsteps + 1):
        v += 0.5 * dt * S.FTM2V * f / m
        x += dt * v
        moved = np.sqrt(((x - x_built) ** 2).sum(axis=1).max())
        assert moved < 0.5 * skin                          # the host list is valid for this step
        if step % rebuild_every == 0 or moved > 0.35 * skin:
            x = S.wrap(s.box, x)                          # Domain::remap + Comm::borders + Neighbor::build
            eng = make_engine(S.System(s.box, x.copy(), s.type, s.tag, s.mass))
            x_built = x.copy()
        o = eng.compute(x, eflag=1, vflag=0)
        f = o["f_owned"]
        v += 0.5 * dt * S.FTM2V * f / m
        if step % every == 0:
            out[step] = (x.copy(), o["eng"] + S.kinetic_energy(m[:, 0], v), o["eng"])
    return out


def _device_run(ctx, style, s, v0, nsteps, every, skin, cutghost, map_):
    d = resident.DeviceDomain(ctx, style, s, cutghost, skin, map_, v0=v0)
    d.compute(1, 0)
    out = {}

    def sample(step):
        t = d.thermo()
        got = ctx.md_download(d.nlocal, want=("x",))
        x = np.zeros((s.n, 3))
        x[d.tags_local - 1] = got["x"]
        out[step] = (x, t["pe"] + t["ke"], t["pe"])

    sample(0)
    for step in range(1, nsteps + 1):
        ev = 1 if step % every == 0 else 0
        d.step(ev, 0, rebuild="auto", defer_final=not ev)
        if ev:
            sample(step)
    return out, d


def _compare(s, host, dev, xtol=1e-8, etol=1e-9):
    worst_x = worst_e = 0.0
    for step in sorted(host):
        xh, eh, _ = host[step]
        xd, ed, _ = dev[step]
        dx = xd - xh
        dx -= np.round(s.box.x2lamda(dx + s.box.lo)) @ s.box.h.T     # same atom, possibly another periodic image
        worst_x = max(worst_x, float(np.abs(dx).max()))
        worst_e = max(worst_e, abs(ed - eh) / s.n)
    assert worst_x < xtol, worst_x
    assert worst_e < etol, worst_e
    return worst_x, worst_e


def test_rebomos_400_steps_from_300K_follow_the_oracle(oracle, monkeypatch):
    """2x2x2 replica of the reference cell (2304 atoms), 300 K, 400 steps: >= 1 style-list build (inner skin 0.5 A) and
    row prunings on the device; host loop with oracle forces"""
    monkeypatch.setenv("MDP_INNER_SKIN", "0.5")
    P = oracle.rebomos_params(POT_REBOMOS)
    s = S.replicate(S.rebomos_bulk_cell(), (2, 2, 2))
    v0 = S.gaussian_velocities(s, 300.0, seed=41)
    nsteps, every, skin = 400, 50, 2.0
    host = _host_run(lambda sy: mdref.RebomosCPU(oracle, P, sy, skin=skin), s, v0, nsteps, every, skin, rebuild_every=100)
    ctx = capi.Context(0)
    p = capi.read_rebomos_file(POT_REBOMOS)
    ctx.rebomos_set_params(p)
    dev, d = _device_run(ctx, capi.STYLE_REBOMOS, s, v0, nsteps, every, skin, 3.0 * p.rcmax[0][0] + skin, [0, 0, 1])
    builds, prunes = ctx.md_neighbor_stats()[7], ctx.md_prune_stats()
    ctx.close()
    assert builds >= 2                                      # the first build + at least one fired by the displacement trigger
    assert prunes["prunings"] >= 2 and prunes["late"] == 0
    _compare(s, host, dev)
    # and the energy does what the oracle's does: its drift is the potential's (unshifted Lennard-Jones cutoff)
    e_host = np.array([host[k][1] for k in sorted(host)])
    e_dev = np.array([dev[k][1] for k in sorted(dev)])
    assert np.abs((e_dev - e_dev[0]) - (e_host - e_host[0])).max() / s.n < 1e-9


def test_aeam_400_steps_from_863K_follow_the_oracle(oracle):
    """4000 atoms of the alloy (8 % Si: angular centres, three-body forces, both table sets), 863 K, 400 steps with the
    deferred on-device `check yes` (several reneighborings: remap, Hilbert re-sorting, new tile lists) and row prunings"""
    T = oracle.aeam_pot(POT_AEAM)
    af = capi.AeamFile(POT_AEAM)
    s = S.fcc_cell(4.045, 10, frac_type2=0.08, seed=51)
    s.mass[1:3] = af.mass[:2]
    v0 = S.gaussian_velocities(s, 863.0, seed=52)
    nsteps, every, skin = 400, 50, 1.0
    host = _host_run(lambda sy: mdref.AeamCPU(oracle, T, sy, skin=skin), s, v0, nsteps, every, skin, rebuild_every=25)
    ctx = capi.Context(0)
    tabs = af.build()
    ctx.aeam_set_tables(tabs)
    dev, d = _device_run(ctx, capi.STYLE_AEAM, s, v0, nsteps, every, skin, float(af.cut_table(tabs).max()) + skin, None)
    prunes = ctx.md_prune_stats()
    ctx.close()
    assert d.builds >= 3 and d.dangerous == 0               # reneighborings fired by the displacement flag, none late
    assert prunes["prunings"] >= 3 and prunes["late"] == 0
    _compare(s, host, dev)
