"""bench.py with two ranks on the one GPU of the test box (--share-gpus: a rehearsal, the ranks share the device): the launcher starts the
ranks itself, every rank processes its own shard on the device and checks it against the oracle, rank 0 prints the one line.  What the
gloo tests on CPU cannot show: that the N > 1 path runs on hardware end to end."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_share_the_gpu_and_print_one_line():
    env = {k: v for k, v in os.environ.items() if not k.startswith("DSPEED_HIP_") and k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpus", "--rows", "60000", "--steps", "3", "--warmup", "1",
                        "--no-cpu"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == [0, 1] and d["scaling"] == "weak"
    assert d["config"]["rows_per_gpu"] == 60000 and d["value"] > 0 and d["parity_max_rel_vs_oracle"] <= d["parity_bar"]
    assert d["roofline"]["kernel_ms_avg_per_rank_min"] <= d["roofline"]["kernel_ms_avg_per_rank_max"]
