"""N > 1 path on CPU: world_size-2 (and 3, uneven) gloo process groups run genfer_amd.dist.sharded_conv
with the oracle injected as the local compute; every rank must end up with the complete product,
bit-identical to the unsharded oracle product."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_oracle_lib, splitmix64_uniform


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_slabs():
    lib = load_oracle_lib()
    szp = ctypes.POINTER(ctypes.c_size_t)
    lib.orc_mul_slabs_timed.restype = ctypes.c_double
    lib.orc_mul_slabs_timed.argtypes = [ctypes.c_void_p, szp, ctypes.c_void_p, szp, ctypes.c_void_p, szp,
                                        ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t,
                                        ctypes.POINTER(ctypes.c_double)]

    def conv_slabs(x, y, z, lo, hi):
        nd = z.dim()
        sz = lambda t: (ctypes.c_size_t * nd)(*t.shape)
        z[lo:hi] = 0.0
        macs = ctypes.c_double()
        lib.orc_mul_slabs_timed(x.data_ptr(), sz(x), y.data_ptr(), sz(y), z.data_ptr(), sz(z), nd, lo, hi,
                                ctypes.byref(macs))

    return conv_slabs


def _worker(rank, world, port, shape, out_dir):
    import sys

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from genfer_amd.dist import broadcast_operands, sharded_conv

    n = int(np.prod(shape))
    if rank == 0:  # operands originate on rank 0 and are replicated by the collective
        x = torch.from_numpy(splitmix64_uniform(1, n).reshape(shape))
        y = torch.from_numpy(splitmix64_uniform(2, n).reshape(shape))
    else:
        x, y = torch.zeros(shape, dtype=torch.float64), torch.zeros(shape, dtype=torch.float64)
    broadcast_operands(x, y, src=0)
    z = torch.full(shape, float("nan"), dtype=torch.float64)
    sharded_conv(x, y, z, _oracle_slabs())
    np.save(os.path.join(out_dir, f"z{rank}.npy"), z.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,shape", [(2, (8, 6, 7)), (2, (7, 5, 6)), (3, (9, 4, 5))])
def test_sharded_conv_gloo(tmp_path, world, shape):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, shape, str(tmp_path)), nprocs=world, join=True)
    n = int(np.prod(shape))
    x = torch.from_numpy(splitmix64_uniform(1, n).reshape(shape))
    y = torch.from_numpy(splitmix64_uniform(2, n).reshape(shape))
    want = torch.zeros(shape, dtype=torch.float64)
    _oracle_slabs()(x, y, want, 0, shape[0])
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"z{r}.npy"))
        assert np.array_equal(got, want.numpy()), f"rank {r} differs"
