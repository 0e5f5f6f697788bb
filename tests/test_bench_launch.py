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


def test_a_crashed_rank_ends_the_launch():
    # rank 1 exits before the rendezvous: the launcher must terminate rank 0 (which would wait for its peer until gloo's timeout) and
    # return rank 1's code, without a benchmark line
    import time

    t0 = time.time()
    r = _run("--gpus", "2", "--dry-run", env={"V1T_BENCH_FAIL_RANK": "1"})
    assert r.returncode == 3, (r.returncode, r.stderr[-1000:])
    assert time.time() - t0 < 120
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_a_hung_rank_ends_the_launch():
    # rank 1 stops making progress before the rendezvous (the launcher polled for CRASHED children only). Its own
    # watchdog (no progress mark for V1T_BENCH_DEADLINE_S) exits 124; the launcher terminates rank 0, which sits in the rendezvous,
    # and returns 124 - no benchmark line, long before gloo's timeout
    import time

    t0 = time.time()
    r = _run("--gpus", "2", "--dry-run", env={"V1T_BENCH_HANG_RANK": "1", "V1T_BENCH_DEADLINE_S": "5", "V1T_DIST_TIMEOUT_S": "100"})
    assert r.returncode == 124, (r.returncode, r.stderr[-1000:])
    assert time.time() - t0 < 90
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())
    # (whichever watchdog fires first ends the launch: the hung rank's, or rank 0's, which waits for it in the rendezvous)
    assert "no progress" in r.stderr and ("rank 1" in r.stderr or "rank 0" in r.stderr)


def test_launch_deadline_terminates_all_ranks():
    # both the per-rank watchdogs out of the picture (long deadline): the launcher's own deadline ends a launch whose ranks all sit still
    import time

    t0 = time.time()
    r = _run("--gpus", "2", "--dry-run", env={"V1T_BENCH_HANG_RANK": "1", "V1T_BENCH_DEADLINE_S": "600", "V1T_BENCH_LAUNCH_DEADLINE_S": "8", "V1T_DIST_TIMEOUT_S": "300"})
    assert r.returncode == 124, (r.returncode, r.stderr[-1000:])
    assert time.time() - t0 < 90
    assert "launch deadline" in r.stderr


def test_an_error_inside_the_run_still_prints_the_line_with_the_failure():
    # an RCCL / HIP error on a rank (here: a test hook that raises inside run_training, before anything touches the GPU): rank 0 prints ONE JSON
    # line whose value is null and whose config.exchange says what failed, and the exit code is not 0
    r = _run("--gpus", "1", "--steps", "1", "--warmup", "0", env={"V1T_BENCH_RAISE": "simulated collective failure"})
    assert r.returncode == 1, (r.returncode, r.stderr[-800:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["value"] is None and d["ms_per_step"] is None and "simulated collective failure" in d["failed"]
    assert d["config"]["exchange"].startswith("failed: ") and d["n_gpus"] == 1 and d["unit"] == "images/s"
    assert "simulated collective failure" in r.stderr  # the traceback goes to stderr


import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_line_contract_on_the_gpu():
    """One short run of the headline configuration: exactly one JSON line, the keys the driver reads, numbers consistent with each
    other (value = images of a step / its time, the dominant kernel's launches inside the timed region, fraction = achieved / peak)."""
    r = _run("--steps", "3", "--warmup", "1", "--min-seconds", "0", "--no-cpu-baseline")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["config"]["step_path"] == "native" and d["config"]["path"] == "native"  # which step ran is part of every line
    assert d["roofline"]["traffic_stale"] in (True, False)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "images/s" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["global_batch"] / d["ms_per_step"] * 1e3) <= 1e-3 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] in ("mfma", "hbm") and rf["unit"] in ("TFLOP/s", "GB/s")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.05 < rf["frac"] < 1.0
    assert rf["launches"] >= 3 * 4  # four blocks per step, every launch of the timed steps measured (hipEvents on the kernel's stream)
    rh = d["roofline_hbm"]  # the second-largest kernel family (HBM-bound): the dQ GEMM over the materialised dS', algorithmic bytes / live duration
    assert rh["bound"] == "hbm" and rh["unit"] == "GB/s" and rh["peak"] == 8000.0 and rh["launches"] >= 3 * 4
    assert abs(rh["frac"] - rh["achieved"] / rh["peak"]) < 1e-3 and 0.2 < rh["frac"] < 1.0
    assert abs(rh["achieved"] - rh["bytes_per_launch"] / (rh["avg_ms"] * 1e-3) / 1e9) <= 1e-3 * rh["achieved"]
    assert rh["traffic"] is None or rh["traffic"] >= 0.9 * rh["bytes_per_launch"]  # counters cannot show less than the algorithmic bytes
    assert 5.0 < d["ms_per_step"] < 200.0
