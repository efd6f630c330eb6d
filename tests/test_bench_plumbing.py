"""bench.py's N-rank path without a GPU: `--plumbing --gpus 8` under gloo on the loopback address -- the launcher (bench.py
spawns its own ranks through torch.distributed.run when WORLD_SIZE is unset, as the driver's `--gpus N` call does), the
rendezvous, shards of unequal size, the data-parallel bucket's ONE all-reduce with B_r / B weights (the Session code the
GPU path runs), barrier-bracketed timing with the max over ranks, rank 0 printing ONE JSON line with the driver's keys."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("ranks", [8, 2])
def test_bench_plumbing_json_contract(ranks):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--plumbing", "--gpus", str(ranks), "--steps", "4",
                          "--warmup", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]            # rank 0 alone prints
    r = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "rccl_ranks", "allreduce_us"):
        assert key in r, key
    assert r["plumbing"] is True and "PLUMBING" in r["metric"]
    assert "all-reduce" in r["value_is"]      # N > 1: the headline carries the collective (the GPU line: the training step)
    assert r["n_gpus"] == ranks and r["rccl_ranks"] == ranks and r["backend"] == "gloo"
    assert r["steps"] == 4 and r["warmup"] == 1 and r["scaling"] == "weak" and r["higher_is_better"] is True
    assert r["reduced_gradient_ok"] is True                # sum_r B_r (r + 1) / sum_r B_r on every rank
    assert r["allreduce_bytes"] == 4 * (31353 + 10) or r["allreduce_bytes"] > 4 * 10
    assert r["value"] > 0 and abs(r["value"] - ranks * 2 * 1e3 / r["ms_per_step"]) / r["value"] < 1e-3
