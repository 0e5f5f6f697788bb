"""bench.py's own launcher: `python bench.py --gpus N` without torchrun must start N ranks (or fail), never
print a single-GPU line. CPU-only: --dry-run joins a gloo group over the launcher's environment."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=300)


def test_gpus2_without_torchrun_spawns_two_ranks():
    r = _run("--gpus", "2", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_in_group"] == 2


def test_gpus_without_enough_devices_fails_loudly():
    # no GPU in the build container (and one on the GPU box): --gpus 64 without torchrun must not print a benchmark line
    r = _run("--gpus", "64", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert r.returncode != 0
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())
    assert "GPU" in r.stderr


def test_world_size_mismatch_is_an_error():
    r = _run("--gpus", "1", "--dry-run", env={"WORLD_SIZE": "2", "RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0
