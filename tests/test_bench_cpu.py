"""bench.py --gpus N from a plain shell, as far as a machine without a GPU can
take it: the script starts its own N ranks (torch.distributed.run on
127.0.0.1) before anything touches a device, the ranks meet over gloo, cut
the box into slabs and check that their X exchange schedules -- the lists the
library hands to RCCL -- pair up. (--dry-run 1: no device call is made.)"""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout           # ONE JSON line on stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 3])
def test_bench_starts_its_own_ranks(n):
    size = ["--size", "12", "8", "8"] if n == 3 else []
    d = _bench("--gpus", str(n), "--dry-run", "1", *size)
    assert d["dry_run"] and d["n_gpus"] == n and d["ranks_met"] == n
    assert d["schedules_pair_up"]
    assert d["config"]["nlocal"][0] * n == (12 if n == 3 else 256)


def test_bench_weak_scaling_config_5():
    d = _bench("--gpus", "4", "--dry-run", "1", "--config", "5")
    assert d["scaling"] == "weak"
    assert d["config"]["nlocal"] == [64, 512, 256]
    assert d["config"]["workload"].startswith("D3Q27 256x512x256")
    # 9 of 27 populations cross a face, two faces, 514 x 258 sites each
    assert d["bytes_per_exchange_per_rank"] == 2 * 9 * 514 * 258 * 8
