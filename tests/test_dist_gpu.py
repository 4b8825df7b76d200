"""SURVEY §8e on the GPU: the output-sharded product with the REAL local compute (HIP kernels), two ranks on one
device with a gloo rendezvous (an 8-GPU RCCL node is only available to the driver).  tests/test_dist_gloo.py covers
the planner and the collectives on CPU with the oracle injected."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_product_two_ranks_one_gpu(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + world), os.path.join(ROOT, "tests", "_dist_gpu_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "DIST_GPU_OK" in out.stdout


@pytest.mark.gpu
def test_bench_n2_control_flow_dry_run_on_one_gpu():
    """bench.py's own N > 1 control flow (rendezvous, agreement all-reduces, sharded step, max-over-ranks timing, the
    per-rank gather, rank 0's single JSON line) executed BEFORE the driver's 8-GPU run does it for the first time: two
    ranks on one device, gloo rendezvous, torch.distributed exchange (RCCL refuses two ranks on one device).  No scaling
    number is taken from this — the line must say that the exchange was not the library's own."""
    import json

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", GFT_BENCH_BACKEND="gloo", GFT_BENCH_EXCHANGE="torch")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29611", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", "small", "--no-clock"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong"
    assert d["exchange"] == "torch" and len(d["per_rank"]) == 2
    assert d["sharded_vs_single_gpu_max_rel_err"] <= 1e-10
    assert all(r["kernel_ms"] > 0 for r in d["per_rank"])
    assert "exchange_fell_back" in d or d.get("exchange_requested") == "torch"
