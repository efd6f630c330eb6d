"""bench.py keeps the driver's contract: one JSON line with the agreed keys, measured on this GPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract(cuda_device):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--cpu-seconds", "2", "--train-steps", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines            # exactly ONE line on stdout
    r = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in r, key
    assert r["n_gpus"] == 1 and r["steps"] == 3 and r["warmup"] == 1 and r["higher_is_better"] is True
    assert r["vs_baseline"] is None and r["scaling"] == "weak" and r["data"] == "synthetic"
    assert "workload" in r["config"] and "n=40" in r["config"]["workload"]
    assert r["value"] > 0 and abs(r["value"] - 32 * 1e3 / r["ms_per_step"]) / r["value"] < 1e-3
    roof = r["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    cpu = r["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] == "port" and cpu["cores"] >= 1
    assert "error" not in (r.get("train") or {})
    # N = 1: the headline is the forward pass (BASELINE's metric as the reference runs it), no guard bit was raised
    assert r["value_is"] == "forward pass" and r["forward_value"] == r["value"] and r["range_guard_bits"] == 0


def test_bench_two_ranks_gloo_on_one_gpu(cuda_device):
    """The N > 1 path of bench.py (rendezvous, barriers, max-over-ranks timing, the training all-reduce, rank 0
    printing alone) launched the way the driver launches it; gloo instead of RCCL so that two ranks can share
    this box's single GPU."""
    env = dict(os.environ, TSPGNN_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--train-steps", "2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["cpu_baseline"] is None
    assert r["config"]["global_batch"] == 2 * r["config"]["per_gpu_batch"]
    assert abs(r["value"] - 2 * 32 * 1e3 / r["ms_per_step"]) / r["value"] < 1e-3      # whole-job aggregate
    assert "error" not in r["train"] and "world 2" in r["train"]["what"]
    # N > 1: the headline IS the training step -- the only step of this path with a collective --
    # timed over exactly --steps steps; the collective-free forward sits in its own field
    assert r["value"] == r["train"]["value"] and r["ms_per_step"] == r["train"]["ms_per_step"]
    assert r["train"]["steps"] == r["steps"] == 3 and r["train"]["rccl_ranks"] == 2
    assert r["value_is"].startswith("training step") and "all-reduce" in r["value_is"]
    assert r["forward_value"] > r["value"] and r["forward_ms_per_step"] < r["ms_per_step"]


def test_bench_spawns_its_own_ranks(cuda_device):
    """`python bench.py --gpus 2` without a launcher (WORLD_SIZE unset) starts one rank per GPU itself -- the way the
    driver calls `--gpus 1` -- and still prints exactly one JSON line; the training block carries the scaling fields."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(TSPGNN_DIST_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--train-steps", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["train"]["n_gpus"] == 2 and r["train"]["scaling"] == "weak"
    assert r["train"]["value"] > 0 and "all-reduce" in r["train"]["collective"]
