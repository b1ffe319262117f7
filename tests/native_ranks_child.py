"""Child process of tests/test_gpu_native_ranks.py: the library's OWN transport (csrc/comm_rccl.hip: grouped
ncclSend/ncclRecv, count all-gathers, the `check yes` word riding in the halo, whole steps in two calls) with 2, 4 and 8
ranks on ONE GPU.  RCCL refuses that, so MDP_RCCL_LIBRARY points the library at the test double of
tests/native/fake_rccl.cpp (which fails where a wrong schedule would hang on the wire); the ranks are threads of this
process, each with its own context -- what a C++ host driving several GPUs from threads does.  A process binds ONE RCCL
object for its lifetime, which is why these cases do not run inside the pytest process (its one-rank tests use RCCL).

usage: native_ranks_child.py OUT.json [case ...]      prints nothing; writes {case: result | {"error": text}}"""
import json
import os
import sys
import time
import traceback

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import conftest  # noqa: E402,F401  (registers the package)
from conftest import POT_AEAM, POT_REBOMOS  # noqa: E402
from lammps_plugins_amd.host import capi, resident, system as S  # noqa: E402

MAP = [0, 0, 1]


def _ctx(style):
    ctx = capi.Context(0)
    if style == "rebomos":
        p = capi.read_rebomos_file(POT_REBOMOS)
        ctx.rebomos_set_params(p)
        return ctx, capi.STYLE_REBOMOS, 3.0 * p.rcmax[0][0] + 2.0, 2.0, MAP
    af = capi.AeamFile(POT_AEAM)
    tabs = af.build()
    ctx.aeam_set_tables(tabs)
    ctx._af = (af, tabs)
    return ctx, capi.STYLE_AEAM, float(af.cut_table(tabs).max()) + 1.0, 1.0, None


def _system(style, pure_metal=False):
    if style == "rebomos":
        s = S.replicate(S.rebomos_bulk_cell(), (3, 3, 2))                  # 5 184 atoms
        v0 = S.gaussian_velocities(s, 300.0, seed=11) + np.array([60.0, -45.0, 30.0])
        return s, v0
    s = S.jitter(S.fcc_cell(4.045, 16, frac_type2=0.0 if pure_metal else 0.03, seed=5), 0.05, seed=6)   # 16 384 atoms
    s.mass[1:3] = capi.AeamFile(POT_AEAM).mass[:2]
    return s, S.gaussian_velocities(s, 863.0, seed=7) + np.array([40.0, 25.0, -30.0])


def _by_tag(dom, want):
    got = dom.ctx.md_download(dom.nlocal, want=want)
    return dom.tags_local, {k: got[k] for k in want}


def _trajectory(world, style, s, v0, steps, native, thermo_at=()):
    """NVE, reneighboring decided by the run itself: one rank reads its deferred on-device flag ("auto"), the native
    ranks the word that travelled with the previous step's halo ("halo")."""

    def rank_fn(r, make_tr):
        ctx, st, cutghost, skin, map_ = _ctx(style)
        tr = make_tr(ctx) if world > 1 else None
        d = resident.DeviceDomain(ctx, st, s, cutghost, skin, map_, v0=v0, transport=tr)
        d.compute(0, 0)
        rows = []
        for step in range(1, steps + 1):
            ev = 1 if (step in thermo_at or step == steps) else 0
            d.step(ev, ev, rebuild="halo" if world > 1 else "auto", defer_final=not ev)
            if ev:
                rows.append(d.thermo())
        tags, a = _by_tag(d, ("x", "v", "f"))
        info = ctx.dd_comm_step_info() if world > 1 else None
        dd = ctx.dd_info()
        prunes = ctx.md_prune_stats()
        aeam = ctx.md_aeam_state() if style == "aeam" else None
        out = dict(tags=tags, x=a["x"], v=a["v"], f=a["f"], rows=rows, builds=d.builds, late=d.dangerous, info=info,
                   nlocal=d.nlocal, nrecv=d.nrecv, nsend=d.nsend, prunings=prunes["prunings"], aeam=aeam,
                   overlapped=d.aeam_overlapped, left_last=dd["left_last"])
        ctx.close()
        return out

    res = [rank_fn(0, None)] if world == 1 else resident.run_ranks(world, rank_fn, native=native)
    n = s.n
    x, v, f, seen = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 3)), np.zeros(n, dtype=int)
    for r in res:
        i = r["tags"] - 1
        x[i], v[i], f[i] = r["x"], r["v"], r["f"]
        seen[i] += 1
    return dict(x=x, v=v, f=f, owned_once=bool(np.all(seen == 1)), ranks=res)


def _compare(s, one, many):
    dx = many["x"] - one["x"]
    dx -= np.round(s.box.x2lamda(dx + s.box.lo)) @ s.box.h.T          # same atom, possibly another periodic image
    r1, rn = one["ranks"][0]["rows"], many["ranks"][0]["rows"]
    return dict(
        owned_once=many["owned_once"], dx=float(np.abs(dx).max()), dv=float(np.abs(many["v"] - one["v"]).max()),
        df=float(np.abs(many["f"] - one["f"]).max()),
        pe_rel=[abs(a["pe"] - b["pe"]) / abs(b["pe"]) for a, b in zip(rn, r1)],
        ke_rel=[abs(a["ke"] - b["ke"]) / max(abs(b["ke"]), 1e-300) for a, b in zip(rn, r1)],
        press_abs=[abs(a["press"] - b["press"]) for a, b in zip(rn, r1)],
        builds_one=one["ranks"][0]["builds"], builds=[r["builds"] for r in many["ranks"]],
        late=[r["late"] for r in many["ranks"]], late_one=one["ranks"][0]["late"],
        reneighbors=[r["info"]["reneighbors"] for r in many["ranks"]],
        nlocal=[r["nlocal"] for r in many["ranks"]], nrecv=[r["nrecv"] for r in many["ranks"]],
        prunings=[r["prunings"] for r in many["ranks"]], overlapped=[r["overlapped"] for r in many["ranks"]],
        ghost_forces=[bool(r["info"]["ghost_forces"]) for r in many["ranks"]],
        policy=[r["info"]["overlap_policy"] for r in many["ranks"]],
        policy_fixed=[r["info"]["overlap_policy_fixed_by_env"] for r in many["ranks"]],
        policy_trial_ms=[r["info"]["overlap_policy_trial_ms"] for r in many["ranks"]],
        interior_tiles=[r["aeam"]["interior_tiles"] if r["aeam"] else None for r in many["ranks"]],
        tiles=[r["aeam"]["tiles"] if r["aeam"] else None for r in many["ranks"]])


_ONE = {}


def _one_rank(style, pure, steps, thermo_at):
    key = (style, pure, steps, tuple(thermo_at))
    if key not in _ONE:
        s, v0 = _system(style, pure)
        _ONE[key] = (s, v0, _trajectory(1, style, s, v0, steps, False, thermo_at))
    return _ONE[key]


def case_steps(style, world, pure=False):
    """hot drifting system: reneighborings (with migration between the bricks) decided by the flag in the halo,
    aeam's fp / ghost-force exchange behind the interior tiles, prunings that send single ranks down the blocking order"""
    steps, thermo_at = (60, (20, 40)) if style == "rebomos" else (48, (16, 32))
    s, v0, one = _one_rank(style, pure, steps, thermo_at)
    many = _trajectory(world, style, s, v0, steps, True, thermo_at)
    return _compare(s, one, many)


def case_fixed_policies():
    """every order of compute against exchanges by itself (MDP_OVERLAP_POLICY): same trajectory"""
    out = {}
    for style, pols in (("rebomos", ("split", "lead", "blocking", "first", "inline")), ("aeam", ("split", "lead", "blocking", "inline"))):
        for pol in pols:
            os.environ["MDP_OVERLAP_POLICY"] = pol
            try:
                out[f"{style}_{pol}"] = case_steps(style, 2)
            finally:
                del os.environ["MDP_OVERLAP_POLICY"]
    return out


def case_library():
    name, double = capi.comm_library()
    return dict(name=name, double=double)


def case_forced_and_blocking_calls(world=2):
    """the piecewise calls of the transport (not the two-call step) between two ranks: forced reneighborings through
    mdp_dd_comm_reneighbor, forward_begin/_end around compute_begin/_end, aeam's blocking forward_scalar / reverse"""
    out = {}
    for style in ("rebomos", "aeam"):
        s, v0 = _system(style)

        def run(w):
            def rank_fn(r, make_tr):
                ctx, st, cutghost, skin, map_ = _ctx(style)
                tr = make_tr(ctx) if w > 1 else None
                d = resident.DeviceDomain(ctx, st, s, cutghost, skin, map_, v0=v0, transport=tr)
                if w > 1:          # the many-call step of resident.DeviceDomain.step instead of the two-call one
                    d.native_two_call = False
                d.compute(1, 1)
                th0 = d.thermo()
                for step in range(1, 13):
                    d.step(0, 0, rebuild=step % 4 == 0)
                d.compute(1, 1)
                th = d.thermo()
                tags, a = _by_tag(d, ("x", "f"))
                ctx.close()
                return dict(tags=tags, x=a["x"], f=a["f"], th0=th0, th=th)

            res = [rank_fn(0, None)] if w == 1 else resident.run_ranks(w, rank_fn, native=True)
            x, f = np.zeros((s.n, 3)), np.zeros((s.n, 3))
            for r in res:
                x[r["tags"] - 1], f[r["tags"] - 1] = r["x"], r["f"]
            return x, f, res[0]["th0"], res[0]["th"]

        x1, f1, a0, a1 = run(1)
        xn, fn, b0, b1 = run(world)
        dx = xn - x1
        dx -= np.round(s.box.x2lamda(dx + s.box.lo)) @ s.box.h.T
        out[style] = dict(dx=float(np.abs(dx).max()), df=float(np.abs(fn - f1).max()),
                          pe0_rel=abs(b0["pe"] - a0["pe"]) / abs(a0["pe"]), pe_rel=abs(b1["pe"] - a1["pe"]) / abs(a1["pe"]))
    return out


def case_mismatched_schedule():
    """one rank issues an exchange its peer does not: on the wire a hang, here an error from the library on BOTH ranks"""
    s, v0 = _system("aeam")

    def rank_fn(r, make_tr):
        ctx, st, cutghost, skin, map_ = _ctx("aeam")
        d = resident.DeviceDomain(ctx, st, s, cutghost, skin, map_, v0=v0, transport=make_tr(ctx))
        d.compute(0, 0)
        t0 = time.time()
        try:
            if r == 0:
                ctx.dd_comm_forward_scalar()        # rank 1 goes straight to the all-reduce below
            ctx.dd_comm_allreduce([1.0], op=0)
            err = None
        except capi.MdpError as e:
            err = str(e)
        ctx.close()
        return dict(err=err, seconds=time.time() - t0)

    return resident.run_ranks(2, rank_fn, native=True)


CASES = {
    "library": case_library,
    "rebomos_2": lambda: case_steps("rebomos", 2), "rebomos_4": lambda: case_steps("rebomos", 4),
    "rebomos_8": lambda: case_steps("rebomos", 8),
    "aeam_2": lambda: case_steps("aeam", 2), "aeam_4": lambda: case_steps("aeam", 4), "aeam_8": lambda: case_steps("aeam", 8),
    "aeam_pure_2": lambda: case_steps("aeam", 2, pure=True), "aeam_pure_4": lambda: case_steps("aeam", 4, pure=True),
    "piecewise_2": case_forced_and_blocking_calls,
    "fixed_policies": case_fixed_policies,
    "mismatch": case_mismatched_schedule,
}


def main():
    out_path, names = sys.argv[1], sys.argv[2:] or list(CASES)
    results = {}
    for name in names:
        t0 = time.time()
        try:
            results[name] = CASES[name]()
        except BaseException as e:  # noqa: BLE001 -- reported per case
            results[name] = {"error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc()}
        print(f"[native ranks] {name}: {time.time() - t0:.1f} s", file=sys.stderr, flush=True)
        with open(out_path, "w") as fh:
            json.dump(results, fh, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))


if __name__ == "__main__":
    main()
