"""bench.py's launcher half on the CPU: `python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset, which is
how a driver that does not use torch.distributed.run starts it) has to become N ranks by itself, before anything touches
a GPU, and the one JSON line has to say so."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra=None, timeout=180):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, cwd=ROOT, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


@pytest.mark.parametrize("n", [2, 3])
def test_bench_starts_its_own_ranks(n):
    r = _run(["--gpus", str(n), "--dry-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout  # one JSON line, from rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == n and rec["ranks_reported"] == n and rec["self_launched"] is True
    assert rec["local_ranks"] == list(range(n))


def test_bench_under_a_launcher_does_not_launch_again():
    """With WORLD_SIZE set (torch.distributed.run's environment) the process is a rank, not a parent."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = _run(["--gpus", "1", "--dry-launch"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                               "MASTER_PORT": str(port)})
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads(r.stdout.strip())
    assert rec["n_gpus"] == 1 and rec["self_launched"] is False


def test_bench_refuses_a_world_that_differs_from_gpus():
    r = _run(["--gpus", "2", "--dry-launch"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_a_failing_rank_fails_the_launch():
    """Without a GPU every rank of a real run exits with an error; the parent must report that, not hang or print a line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a host without a GPU")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-pmc"])
    assert r.returncode != 0
    assert r.stdout.strip() == ""


def test_reduce_cadence_rule():
    sys.path.insert(0, ROOT)
    import bench
    n1024 = 1024 * 1024 * 3
    n4k = 3840 * 2160 * 3
    assert bench.reduce_cadence(2.3, n1024, 1) == 1
    assert bench.reduce_cadence(2.3, n1024, 8) >= 4      # a 12.6 MB reduce against a 2.3-ms frame: several frames per reduce
    assert bench.reduce_cadence(48.0, n4k, 8) <= 2       # 99.5 MB against 48 ms: every frame or every other
    assert bench.reduce_cadence(0.0, n4k, 8) == 16
