"""Output-sharded product across the GPUs of one node: one process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Every output slab ``z[k0, ...]`` of the truncated product depends only on ``x[0..=k0]`` and
``y[0..=k0]`` and on no other output (src/multivariate_taylor.rs:1001-1011), so the path shards
without any reduction: operands are replicated (``broadcast_operands`` when they originate on one
rank), every rank computes its own leading-axis slabs — assigned by the folded plan of
``gft_plan_slabs`` so that the triangular work is balanced — and the result slabs are exchanged with
two all-gathers (low groups, mirrored high groups); the first one is started asynchronously as soon as its
group is computed and overlaps the second group's kernels.  When the leading axis does not divide evenly
the fallback is an all-reduce of zero-initialised tensors (adding zeros is exact, so the result is
bit-identical to the single-GPU one).

The local compute is injected (``conv_slabs(x, y, z, lo, hi)``): the product uses
``genfer_amd.conv_raw`` (HIP kernels); the CPU tests inject the oracle to exercise the planner and
the collectives under gloo.
"""
from __future__ import annotations

from typing import Callable, Sequence

import torch
import torch.distributed as dist

from . import plan_slabs


def broadcast_operands(x: torch.Tensor, y: torch.Tensor, src: int = 0, group=None) -> None:
    """Replicate the operands from ``src`` (RCCL broadcast over xGMI: the root pushes to its peers
    on all links concurrently; 2 x 134 MB at 64^4 is ~2 ms, <= 3 % of the product)."""
    dist.broadcast(x, src=src, group=group)
    dist.broadcast(y, src=src, group=group)


def local_ranges(n0: int, world: int, rank: int):
    (a, b), (c, d), even = plan_slabs(n0, world, rank)
    launches = [(a, d)] if b == c else [r for r in ((a, b), (c, d)) if r[1] > r[0]]
    return (a, b), (c, d), even, launches


def sharded_conv(
    x: torch.Tensor,
    y: torch.Tensor,
    z: torch.Tensor,
    conv_slabs: Callable[[torch.Tensor, torch.Tensor, torch.Tensor, int, int], None],
    group=None,
    before_local: Callable[[], None] | None = None,
    after_local: Callable[[], None] | None = None,
) -> torch.Tensor:
    """z = x (*) y with the leading output axis sharded over the process group.  ``x`` and ``y``
    must already hold identical data on every rank; ``z`` (full result shape, same on every rank) is
    filled completely on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n0 = z.shape[0]
    g0, g1, even, launches = local_ranges(n0, world, rank)
    if world > 1 and not even:
        z.zero_()
    if before_local:
        before_local()
    works = []
    if world > 1 and even:
        plans = [plan_slabs(n0, world, r) for r in range(world)]
        pending = [(0, g0), (1, g1)]  # slab groups whose exchange has not been started yet
    for lo, hi in launches:
        conv_slabs(x, y, z, lo, hi)
        if world > 1 and even:
            # start the exchange of every group this launch completed while the next group is still computing: the
            # collective (on the backend's own stream, ordered after the launch) writes the OTHER ranks' slabs only
            while pending and pending[0][1][1] <= hi and pending[0][1][0] >= lo:
                g, mine = pending.pop(0)
                outs = [z[plans[r][g][0]:plans[r][g][1]] for r in range(world)]
                works.append(dist.all_gather(outs, z[mine[0]:mine[1]], group=group, async_op=True))
    if after_local:  # e.g. record a HIP event: kernel time is reported separately from the exchange
        after_local()
    if world == 1:
        return z
    if even:
        while pending:  # (not reached: every group lies inside one launch)
            g, mine = pending.pop(0)
            outs = [z[plans[r][g][0]:plans[r][g][1]] for r in range(world)]
            works.append(dist.all_gather(outs, z[mine[0]:mine[1]], group=group, async_op=True))
        for w in works:
            w.wait()
    else:
        dist.all_reduce(z, group=group)
    return z


def gpu_conv_slabs(x: torch.Tensor, y: torch.Tensor, z: torch.Tensor, lo: int, hi: int) -> None:
    """The product's local compute: HIP kernels on the caller's tensors (no CPU path).

    Stream contract: torch's fills / uploads of x, y, z and the collectives of ``sharded_conv`` are ordered on
    torch's CURRENT stream, the library launches on its own stream unless told otherwise — so the library adopts
    torch's current stream here (``gft_set_stream`` synchronises the old one once when the stream changes)."""
    import ctypes

    from . import conv_raw, lib

    L = lib()
    cur = torch.cuda.current_stream().cuda_stream
    args = (x.data_ptr(), tuple(x.shape), y.data_ptr(), tuple(y.shape), z.data_ptr(), tuple(z.shape), lo, hi)
    if not cur:
        # torch's legacy default stream (handle 0) cannot be adopted (0 means "the library's own stream"): order the
        # launch with explicit synchronisation on both sides instead — correct, just not asynchronous
        torch.cuda.current_stream().synchronize()
        conv_raw(*args)
        L.gft_synchronize()
        return
    if (L.gft_get_stream() or 0) != cur:
        if L.gft_set_stream(ctypes.c_void_p(cur)) != 0:
            raise RuntimeError((L.gft_last_error() or b"gft_set_stream failed").decode())
    conv_raw(*args)
