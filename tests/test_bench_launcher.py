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
