"""Multi-GPU behind the C ABI (include/gftaylor.h `gft_dist_*`).  RCCL needs one GPU per rank, so what can
be checked without an 8-GPU node is (a) the exchange plan — pure integer arithmetic, simulated here for every rank of
every world size: the in-place all-gather of the low groups and the point-to-point exchange of the mirrored high groups
deliver every slab to every rank exactly once — and (b) on the 1-GPU box, the real RCCL plumbing at world size 1
(dlopen, unique id, ncclCommInitRank, ncclCommCount, the sharded entry point)."""
import ctypes

import numpy as np
import pytest


def _plan(n0, world, rank):
    import genfer_amd

    (a, b), (c, d), even = genfer_amd.plan_slabs(n0, world, rank)
    return (a, b, c, d), even


@pytest.mark.parametrize("n0,world", [(64, 8), (128, 8), (64, 2), (64, 4), (48, 3), (16, 8), (128, 1)])
def test_exchange_plan_delivers_every_slab_once(n0, world):
    """Mirror of dist_exchange (gft_api.hip) for an even split: after the in-place all-gather of the low groups
    (rank r's group must sit at offset r * b0 of the low half) and the grouped sends / receives of the high groups,
    every rank holds every slab, each written exactly once."""
    plans = [_plan(n0, world, r) for r in range(world)]
    assert all(even for _, even in plans)
    for me in range(world):
        have = np.zeros(n0, dtype=int)
        (a, b, c, d), _ = plans[me]
        have[a:b] += 1
        have[c:d] += 1
        b0 = b - a
        for r in range(world):  # ncclAllGather(send = z + mine.lo, recv = z, count = b0): rank r's data lands at r * b0
            (ra, rb, _, _), _ = plans[r]
            assert rb - ra == b0 and ra == r * b0, "low groups must be equal-sized and in rank order for an in-place all-gather"
            if r != me:
                have[r * b0:(r + 1) * b0] += 1
        for peer in range(world):  # ncclRecv(z + theirs.lo, count = theirs.hi - theirs.lo, peer)
            if peer == me:
                continue
            (_, _, pc, pd), _ = plans[peer]
            assert pd - pc == d - c, "send and receive counts of a pair must match"
            have[pc:pd] += 1
        assert (have == 1).all(), (me, have)


def test_uneven_split_is_flagged_for_the_allreduce_fallback():
    _, even = _plan(100, 8, 0)
    assert not even
    covered = np.zeros(100, dtype=int)
    for r in range(8):
        (a, b, c, d), _ = _plan(100, 8, r)
        covered[a:b] += 1
        covered[c:d] += 1
    assert (covered == 1).all()  # disjoint cover: summing zero-filled tensors is exact


@pytest.mark.gpu
def test_rccl_plumbing_world_1_and_sharded_entry_point():
    import torch

    import genfer_amd

    genfer_amd.init(0)
    L = genfer_amd.lib()
    assert L.gft_dist_comm_count() in (0, 1)
    if L.gft_dist_comm_count() == 0:
        genfer_amd.dist_init(0, 1, genfer_amd.dist_unique_id())
    assert L.gft_dist_world() == 1 and L.gft_dist_rank() == 0 and L.gft_dist_comm_count() == 1
    rng = np.random.default_rng(3)
    for xs, ys, zs in [((20, 18, 24), (20, 18, 24), (20, 18, 24)), ((9, 40, 33), (12, 7, 40), (16, 40, 40)), ((150,), (150,), (150,))]:
        x, y = torch.from_numpy(rng.random(xs)).cuda(), torch.from_numpy(rng.random(ys)).cuda()
        z1 = torch.full(zs, float("nan"), dtype=torch.float64, device="cuda")
        z2 = torch.full(zs, float("nan"), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        genfer_amd.conv_raw(x.data_ptr(), xs, y.data_ptr(), ys, z1.data_ptr(), zs)
        genfer_amd.conv_raw_sharded(x.data_ptr(), xs, y.data_ptr(), ys, z2.data_ptr(), zs)
        L.gft_synchronize()
        assert bool(torch.equal(z1, z2))
    buf = torch.arange(16, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    assert L.gft_dist_broadcast(ctypes.c_void_p(buf.data_ptr()), 16, 0) == 0
    L.gft_synchronize()
    assert bool(torch.equal(buf.cpu(), torch.arange(16, dtype=torch.float64)))
    # the self-test every rank runs before a multi-GPU job is timed (bench.py, GFT_DIST_SELFTEST=1): both sharded entries,
    # even and uneven split, bit for bit against the rank's own full product; conv mode and dist_min_macs are restored
    assert L.gft_dist_selftest() == 0, L.gft_last_error()
    # exchange-only event slot: start, local kernels done, exchange done
    assert L.gft_set_option(b"dist_event_slot", 30.0) == 0
    x = torch.from_numpy(rng.random((16, 12, 12))).cuda()
    z = torch.zeros((16, 12, 12), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    genfer_amd.conv_raw_sharded(x.data_ptr(), (16, 12, 12), x.data_ptr(), (16, 12, 12), z.data_ptr(), (16, 12, 12))
    L.gft_synchronize()
    assert L.gft_event_elapsed_ms(30, 31) >= 0.0 and L.gft_event_elapsed_ms(31, 32) >= 0.0
    L.gft_set_option(b"dist_event_slot", -1.0)
