"""GPU: the DEFAULT multi-GPU transport (csrc/comm_rccl.hip: the library's grouped ncclSend/ncclRecv, the count
all-gathers, the `neigh_modify check yes` word riding in the halo, whole steps in mdp_dd_comm_step_begin/_end) with 2, 4
and 8 ranks whose peers are NOT the rank itself -- on one GPU, through the RCCL test double of tests/native/fake_rccl.cpp
(MDP_RCCL_LIBRARY), which fails where a wrong schedule would hang on the wire.  The rank threads run in ONE child process
(tests/native_ranks_child.py: a process binds one RCCL object for its lifetime, and this pytest process uses RCCL itself
in its one-rank tests); `python bench.py --gpus 2` runs as rank PROCESSES on the double.  What 4 real ranks give the
reference: log.rebomos-bulk.4:22,54-56,65-67."""
import json
import os
import subprocess
import sys

import pytest

from lammps_plugins_amd.host import capi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "native_ranks_child.py")


def _env(timeout="30"):
    if not os.path.exists(capi.FAKE_RCCL):
        subprocess.run(["make", "-C", os.path.join(ROOT, "lammps-plugins_amd"), "rccl-double"], check=True)
    return dict(os.environ, MDP_RCCL_LIBRARY=capi.FAKE_RCCL, MDP_FAKE_RCCL_TIMEOUT_S=timeout)


@pytest.fixture(scope="module")
def results(tmp_path_factory):
    out = tmp_path_factory.mktemp("native_ranks") / "results.json"
    log = os.path.join(ROOT, "gpurun_out", "native_ranks_child.log")
    os.makedirs(os.path.dirname(log), exist_ok=True)
    with open(log, "w") as fh:
        p = subprocess.run([sys.executable, CHILD, str(out)], env=_env(), stdout=fh, stderr=subprocess.STDOUT, timeout=1500)
    assert out.exists(), f"the child wrote nothing (exit {p.returncode}); see {log}"
    import shutil
    shutil.copy(out, os.path.join(ROOT, "gpurun_out", "native_ranks_results.json"))
    return json.load(open(out))


def _case(results, name):
    assert name in results, f"case {name} did not run"
    r = results[name]
    assert "error" not in r, r.get("trace", r["error"])
    return r


def test_the_child_is_bound_to_the_double_and_says_so(results):
    r = _case(results, "library")
    assert r["double"] and r["name"].endswith("libfake_rccl.so")


@pytest.mark.parametrize("style,world", [("rebomos", 2), ("rebomos", 4), ("rebomos", 8), ("aeam", 2), ("aeam", 4), ("aeam", 8)])
def test_whole_steps_on_n_ranks_follow_the_one_rank_trajectory(results, style, world):
    """hot drifting systems: the ranks reneighbor (and migrate atoms) by the word that rode in the previous step's halo,
    all on the same step; positions 1e-8 A, thermo rows of the steps in between equal to the one-rank run's"""
    r = _case(results, f"{style}_{world}")
    assert r["owned_once"]
    assert r["dx"] < 1e-8 and r["dv"] < 1e-7 and r["df"] < 1e-7
    assert max(r["pe_rel"]) < 1e-10 and max(r["ke_rel"]) < 1e-9
    assert len(set(r["builds"])) == 1                                   # every rank took the same decisions
    assert r["builds"][0] >= 3 and abs(r["builds"][0] - r["builds_one"]) <= 1
    assert r["reneighbors"] == r["builds"]                              # (the library's count includes the setup's)
    assert r["late"] == [0] * world and r["late_one"] == 0
    assert min(r["nrecv"]) > 0                                          # every rank has REMOTE ghosts
    # the first steps were the library's overlap-policy trial (every order in turn); all ranks kept the same one
    assert len(set(r["policy"])) == 1 and r["policy"][0] in ("split", "lead", "blocking", "first", "inline")
    assert not any(r["policy_fixed"]) and min(r["policy_trial_ms"]) > 0.0
    if style == "aeam":
        assert all(r["ghost_forces"])                                   # 3 % angular atoms: some sit in a shell
        assert all(0 < o for o in r["overlapped"])                      # steps on the phased order ...
        assert sum(r["prunings"]) > world                               # ... and prunings, which are rank-local:
        assert len(set(r["overlapped"])) > 1 or max(r["overlapped"]) < 48   # single ranks took the blocking order


@pytest.mark.parametrize("world", [2, 4])
def test_pure_metal_bricks_skip_the_reverse_exchange_on_every_rank(results, world):
    r = _case(results, f"aeam_pure_{world}")
    assert r["owned_once"] and r["dx"] < 1e-8 and r["df"] < 1e-7
    assert not any(r["ghost_forces"])
    assert max(r["pe_rel"]) < 1e-10
    assert all(0 < o < 48 for o in r["overlapped"])                     # blocking steps among phased ones


def test_every_overlap_policy_by_itself_gives_the_trajectory(results):
    """MDP_OVERLAP_POLICY fixes the order of compute against exchanges (no trial): split / lead / blocking / first / inline"""
    r = _case(results, "fixed_policies")
    assert len(r) == 9
    for name, c in r.items():
        assert c["owned_once"] and c["dx"] < 1e-8 and c["df"] < 1e-7, name
        assert max(c["pe_rel"]) < 1e-10, name
        assert set(c["policy"]) == {name.split("_")[1]} and all(c["policy_fixed"]), name
        if name in ("aeam_blocking", "aeam_inline"):
            assert c["overlapped"] == [0, 0]                            # no step on the phased order
        elif name.startswith("aeam"):
            assert min(c["overlapped"]) > 0


def test_piecewise_transport_calls_between_two_ranks(results):
    r = _case(results, "piecewise_2")
    for style in ("rebomos", "aeam"):
        assert r[style]["dx"] < 1e-8 and r[style]["df"] < 1e-7
        assert r[style]["pe0_rel"] < 1e-11 and r[style]["pe_rel"] < 1e-10


def test_a_schedule_that_differs_between_ranks_is_an_error_not_a_pass(results):
    r = _case(results, "mismatch")
    assert all(x["err"] for x in r), r                                  # both ranks fail ...
    assert any("fake-rccl" in x["err"] for x in r)
    assert max(x["seconds"] for x in r) < 60                            # ... within the double's timeout


@pytest.mark.timeout(900)
@pytest.mark.parametrize("workload,rep,world", [("rebomos", (6, 6, 4), 2), ("aeam", (24, 24, 24), 4)])
def test_bench_line_from_rank_processes_on_the_double(workload, rep, world):
    """`python bench.py --gpus N` exactly as the driver starts it (own rank processes, library transport, two-call
    steps), N ranks sharing this GPU through the double: one line, rccl_ranks = N, marked as a rehearsal"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "30", "--warmup", "5",
           "--workload", workload, "--replicate", *map(str, rep), "--temp", "300", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=_env("60"), capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == world and line["config"]["rccl_ranks"] == world
    assert "TEST DOUBLE" in line["config"]["transport"]
    assert line["config"]["rccl_library_is_test_double"] is True
    assert len(line["config"]["nlocal_per_rank"]) == world and min(line["config"]["remote_ghosts_per_rank"]) > 0
    assert line["value"] > 0


@pytest.mark.timeout(900)
def test_failing_library_transport_falls_back_to_torch_and_says_so():
    """the launcher's one fallback: the library-transport ranks give up (MDP_BENCH_TEST_FAIL_NATIVE), fresh rank processes
    run the torch.distributed transport (gloo-staged here: one GPU), and the line carries `transport_fallback`"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2",
           "--replicate", "6", "6", "4", "--temp", "300", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=dict(_env("60"), MDP_BENCH_TEST_FAIL_NATIVE="1"), capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert "library transport" in line["transport_fallback"] and "exit code" in line["transport_fallback"]
    assert line["n_gpus"] == 2 and "staged" in line["config"]["transport"]
    assert line["config"]["rccl_library_is_test_double"] is False


@pytest.mark.timeout(900)
def test_bench_on_n_ranks_also_runs_the_two_cpp_hosts_on_the_same_ranks():
    """a plain REBO-MoS line on N ranks carries `secondary`: `ddhost -ranks N` (the C-ABI alone) and `minilmp -np N` with
    fix nve/mdp on the library's bricks (the plugin surface), started by rank 0 behind the timed region"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
           "--replicate", "6", "6", "4", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=_env("60"), capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    sec = line["secondary"]
    assert sec["cpp_host_resident_ranks"]["ranks"] == 2 and sec["cpp_host_resident_ranks"]["test_double"] is True
    pl = sec["plugin_load_nve_mdp_ranks"]
    assert pl["ranks"] == 2 and pl["bricks"] == 2 and pl["atoms"] == 288 * 6 * 6 * 4 and pl["steps"] == 200
    # the same system, started from rest, in three hosts: potential energy per atom of the first thermo row
    assert pl["thermo_rows"][0][3] / pl["atoms"] == pytest.approx(line["config"]["pe_per_atom_start_eV"], abs=2e-6)
