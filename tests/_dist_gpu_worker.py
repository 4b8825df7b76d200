"""Worker of tests/test_dist_gpu.py: one process per rank (gloo rendezvous, every rank on cuda:0 of a 1-GPU box),
the output-sharded product through the HIP kernels, checked against the single-GPU product on every rank."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import genfer_amd  # noqa: E402
from genfer_amd.dist import gpu_conv_slabs, sharded_conv  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    genfer_amd.init(0)
    L = genfer_amd.lib()
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    import ctypes

    L.gft_set_stream(ctypes.c_void_p(stream.cuda_stream))
    worst = 0.0
    # even folded split (two all-gathers, the first overlapped), uneven split (all-reduce fallback), rank 4, rank 2
    # through the inner split of the tiled kernel
    for shape in [(32, 24, 40), (30, 20, 24), (16, 12, 10, 16), (96, 200)]:
        rng = np.random.default_rng(7)  # identical operands on every rank
        x = torch.from_numpy(rng.random(shape)).cuda()
        y = torch.from_numpy(rng.random(shape)).cuda()
        z = torch.full(shape, float("nan"), dtype=torch.float64, device="cuda")
        sharded_conv(x, y, z, gpu_conv_slabs)
        ref = torch.zeros(shape, dtype=torch.float64, device="cuda")
        genfer_amd.conv_raw(x.data_ptr(), shape, y.data_ptr(), shape, ref.data_ptr(), shape)
        L.gft_synchronize()
        torch.cuda.synchronize()
        err = float(((z - ref).abs() / ref.abs()).max())
        worst = max(worst, err)
        assert err <= 1e-10, (shape, err)
    dist.barrier()
    if rank == 0:
        print(f"DIST_GPU_OK worst_rel_err={worst:.3e}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
