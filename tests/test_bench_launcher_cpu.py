"""`python bench.py --gpus N` must start its own N ranks (one process per GPU) before anything touches the device, relay rank 0's
line and fail when a rank fails.  Driven here with --dry-run workers (no library, no device): launcher, rendezvous (gloo on
127.0.0.1), shard bookkeeping."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*flags, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *flags], capture_output=True, text=True, timeout=280, env=e)


@pytest.mark.timeout(300)
def test_plain_invocation_starts_n_distinct_ranks():
    r = _run("--gpus", "4", "--dry-run")
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # ONE line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 4 and line["ranks_seen"] == [0, 1, 2, 3]
    assert len(set(line["pids"])) == 4 and line["launcher_pid"] not in line["pids"]  # fresh processes, not the launcher re-executed
    assert line["rows_per_gpu"] == 1_250_000 and line["first_rows"] == [0, 1_250_000, 2_500_000, 3_750_000]  # configs[3] shards


@pytest.mark.timeout(300)
def test_a_failing_rank_fails_the_run_and_prints_no_line():
    r = _run("--gpus", "2", "--dry-run", "--fail-rank", "1")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_flag_and_launcher_must_agree():
    r = _run("--gpus", "8", "--dry-run", env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode == 2 and "must agree" in r.stderr  # never a silent 1-GPU number under an 8-GPU flag


def test_kernel_switches_are_refused():
    r = _run("--dry-run", env={"DSPEED_HIP_ABLATE": "4"})
    assert r.returncode == 4 and "DSPEED_HIP_ABLATE" in r.stderr
