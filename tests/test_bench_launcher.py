"""bench.py --gpus N must never measure another rank count than the one asked for: with WORLD_SIZE unset it starts
its own N rank processes, and it exits non-zero -- without a result line -- when the ranks cannot run (here: no GPU
in this container) or when the launcher's WORLD_SIZE disagrees with --gpus."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_world_size_that_disagrees_with_gpus_is_refused():
    r = _bench(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "refusing" in r.stderr and "{" not in r.stdout


def test_ranks_that_cannot_come_up_give_no_result():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU: there the rank processes must fail")
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r.returncode != 0                       # never falls back to one rank
    assert '"metric"' not in r.stdout
    assert "starting 2 ranks" in r.stderr
    # the one fallback (fresh ranks on the torch.distributed transport) was tried and failed as well: still no result
    assert "starting fresh ranks on the torch.distributed transport" in r.stderr
    assert r.stderr.count("starting 2 ranks") == 2


import pytest


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_n_rank_code_path_over_rccl_prints_one_line():
    """MDP_BENCH_SELF_REMOTE=1 under torch.distributed.run with one rank: bench.py takes its N>1 code path (process
    group on RCCL, collective checks, per-rank gathers; periodic images travel through the all-to-all to the rank
    itself).  stdout must hold the result line and nothing else -- RCCL prints a version banner to stdout when a
    communicator comes up -- and the line must describe the decomposition."""
    import json
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MDP_BENCH_SELF_REMOTE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--replicate",
                        "6", "6", "6", "--temp", "300", "--steps", "20", "--warmup", "3", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    c = d["config"]
    assert d["n_gpus"] == 1 and c["rccl_ranks"] == 1 and "rccl" in c["transport"] and "rank itself" in c["transport"]
    assert c["nlocal_per_rank"] == [62208] and c["remote_ghosts_per_rank"][0] > 0
    assert d["value"] > 0 and d["roofline"]["frac"] > 0
