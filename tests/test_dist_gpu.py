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
