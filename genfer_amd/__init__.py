"""genfer_amd — MI355X-native multivariate-Taylor arithmetic core for Genfer.

The package is a thin host-side mirror of the reference's ``TaylorPoly<F64>`` /
``TaylorPoly<Interval<F64>>`` surface (src/multivariate_taylor.rs) over the C ABI of
``include/gftaylor.h`` implemented by ``genfer_amd/csrc/libgftaylor.so`` (hand-written HIP
kernels for gfx950).  There is no CPU fallback: importing the handle classes without the built
library, or using them without a gfx950 device, fails loudly.

    >>> from genfer_amd import TaylorPoly
    >>> f = TaylorPoly.taylor([[1.0, 2.0], [3.0, 4.0]])
    >>> (f * f).array()
"""
from __future__ import annotations

import ctypes
import os

# Launch-bound programs (10^5 small dependent kernels) run measurably faster with kernel arguments in device memory; the
# HIP runtime reads the flag when it initialises, so it is set as early as this package can (a user's own value wins).
# This is the HOST's choice: libgftaylor itself never touches the environment.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from .taylor import USIZE_MAX, TaylorError, bind  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgftaylor.so")

_lib = None
_classes = {}


def lib() -> ctypes.CDLL:
    """The loaded C-ABI library (raises if it has not been built: see __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing — build it with `make -C genfer_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
            )
        # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.7 /
        # libhsa-runtime64 and load them by path.  If libgftaylor pulled in /opt/rocm's copy first,
        # a later `import torch` would bring up a second runtime that cannot open the GPU
        # ("No HIP GPUs are available").  Importing torch first makes our DT_NEEDED resolve (by
        # soname) to the copy torch already loaded.  torch is plumbing here (streams,
        # torch.distributed), never a compute fallback; without torch the system runtime is used.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = ctypes.CDLL(LIB_PATH)
        _declare_runtime(_lib)
    return _lib


def _declare_runtime(L):
    c = ctypes
    sz = c.POINTER(c.c_size_t)
    dp = c.POINTER(c.c_double)
    L.gft_init.restype, L.gft_init.argtypes = c.c_int, [c.c_int]
    L.gft_shutdown.restype, L.gft_shutdown.argtypes = None, []
    L.gft_set_stream.restype, L.gft_set_stream.argtypes = c.c_int, [c.c_void_p]
    L.gft_get_stream.restype, L.gft_get_stream.argtypes = c.c_void_p, []
    L.gft_synchronize.restype, L.gft_synchronize.argtypes = c.c_int, []
    L.gft_last_error.restype, L.gft_last_error.argtypes = c.c_char_p, []
    L.gft_pool_stats.restype, L.gft_pool_stats.argtypes = None, [sz]
    L.gft_op_stats.restype, L.gft_op_stats.argtypes = None, [sz]
    L.gft_op_stats_ex.restype, L.gft_op_stats_ex.argtypes = c.c_size_t, [sz, c.c_size_t]
    L.gft_event_record.restype, L.gft_event_record.argtypes = c.c_int, [c.c_int]
    L.gft_event_elapsed_ms.restype, L.gft_event_elapsed_ms.argtypes = c.c_float, [c.c_int, c.c_int]
    L.gft_set_conv_mode.restype, L.gft_set_conv_mode.argtypes = c.c_int, [c.c_int]
    L.gft_set_option.restype, L.gft_set_option.argtypes = c.c_int, [c.c_char_p, c.c_double]
    L.gft_conv_raw.restype = c.c_int
    L.gft_conv_raw.argtypes = [c.c_void_p, sz, c.c_void_p, sz, c.c_void_p, sz, c.c_size_t, c.c_size_t, c.c_size_t, c.c_int]
    L.gft_conv_macs.restype = c.c_double
    L.gft_conv_macs.argtypes = [sz, sz, sz, c.c_size_t, c.c_size_t, c.c_size_t]
    L.gft_plan_slabs.restype = c.c_int
    L.gft_plan_slabs.argtypes = [c.c_size_t, c.c_int, c.c_int, sz]
    L.gft_dist_unique_id.restype, L.gft_dist_unique_id.argtypes = c.c_int, [c.c_void_p]
    L.gft_dist_init.restype, L.gft_dist_init.argtypes = c.c_int, [c.c_int, c.c_int, c.c_void_p]
    L.gft_dist_world.restype, L.gft_dist_world.argtypes = c.c_int, []
    L.gft_dist_rank.restype, L.gft_dist_rank.argtypes = c.c_int, []
    L.gft_dist_comm_count.restype, L.gft_dist_comm_count.argtypes = c.c_int, []
    L.gft_dist_shutdown.restype, L.gft_dist_shutdown.argtypes = c.c_int, []
    L.gft_dist_selftest.restype, L.gft_dist_selftest.argtypes = c.c_int, []
    L.gft_dist_broadcast.restype, L.gft_dist_broadcast.argtypes = c.c_int, [c.c_void_p, c.c_size_t, c.c_int]
    L.gft_conv_raw_sharded.restype = c.c_int
    L.gft_conv_raw_sharded.argtypes = [c.c_void_p, sz, c.c_void_p, sz, c.c_void_p, sz, c.c_size_t]


def init(device: int = -1) -> None:
    """Select the HIP device (default: LOCAL_RANK or 0) and create stream + memory pool."""
    L = lib()
    if L.gft_init(device) != 0:
        raise TaylorError((L.gft_last_error() or b"gft_init failed").decode())


def _cls(prefix: str):
    if prefix not in _classes:
        _classes[prefix] = bind(lib(), prefix)
    return _classes[prefix]


def __getattr__(name):  # lazy: `from genfer_amd import TaylorPoly` loads the library on first use
    if name == "TaylorPoly":
        return _cls("gft_")
    if name == "IntervalTaylorPoly":
        return _cls("gfti_")
    raise AttributeError(name)


HOST_LIB_PATH = os.path.join(_HERE, "csrc", "host", "libgfhost.so")
_host = None


def host_lib() -> ctypes.CDLL:
    """The host interpreter (SGCL parser -> GF DAG -> eval -> report), backend-agnostic over the C ABI."""
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise ImportError(f"{HOST_LIB_PATH} is missing — build it with `make -C genfer_amd/csrc/host`")
        _host = ctypes.CDLL(HOST_LIB_PATH)
        _host.gfh_run.restype = ctypes.c_int
        _host.gfh_run.argtypes = [ctypes.c_char_p] * 4 + [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p)]
        _host.gfh_free.argtypes = [ctypes.c_void_p]
    return _host


def run_sgcl_with_backend(source: str, flags: str, backend_lib: str, prefix: str):
    """Run an SGCL program end to end against an explicit TaylorPoly backend library.
    Returns (rc, report_text, timings_dict).  The product entry point is :func:`run_sgcl`."""
    import json

    H = host_lib()
    out, tj = ctypes.c_void_p(), ctypes.c_void_p()
    rc = H.gfh_run(source.encode(), flags.encode(), backend_lib.encode(), prefix.encode(), ctypes.byref(out), ctypes.byref(tj))
    text = ctypes.string_at(out).decode() if out else ""
    timings = json.loads(ctypes.string_at(tj).decode()) if tj else None
    if out:
        H.gfh_free(out)
    if tj:
        H.gfh_free(tj)
    return rc, text, timings


def run_sgcl(source: str, flags: str = ""):
    """`genfer file.sgcl <flags>` on the GPU: the reference's report text (src/main.rs) with every
    TaylorPoly operation executed by libgftaylor's HIP kernels.  `--bounds` selects interval tensors."""
    lib()  # make sure the runtime (and torch's HIP runtime, see lib()) is loaded first
    prefix = "gfti_" if any(t in ("-b", "--bounds") for t in flags.split()) else "gft_"
    rc, text, timings = run_sgcl_with_backend(source, flags, LIB_PATH, prefix)
    if rc != 0:
        raise TaylorError(text)
    return text, timings


OP_STATS = ("linear_scans", "scalar_readbacks", "coefficient_readbacks", "tiled", "staged", "per_output", "host_tier_ops",
            "host_to_device_mirrors")
OP_STATS_EX = ("launches", "deferred_ops", "chains_materialised", "chain_addsub_launches", "launches_in_place", "shallow_products",
               "fused_horner_steps", "unused_7", "unused_8", "riders", "fused_observe_adds", "scans_proven", "nested_adds",
               "graph_executions", "graph_recordings", "batch_launches", "batch_items", "graph_us")


def pool_stats() -> dict:
    """Device memory the library holds, in bytes: pool blocks in use / cached, and the peak (gft_pool_stats; the kernels'
    grow-only workspaces are included in `in_use` and `peak`)."""
    a = (ctypes.c_size_t * 3)()
    lib().gft_pool_stats(a)
    return {"in_use": int(a[0]), "cached": int(a[1]), "peak": int(a[2])}


def op_stats() -> dict:
    """Cumulative counters of the library since gft_init (gft_op_stats + gft_op_stats_ex) by name."""
    L = lib()
    a, b = (ctypes.c_size_t * 8)(), (ctypes.c_size_t * 32)()
    L.gft_op_stats(a)
    n = min(L.gft_op_stats_ex(b, 32), 32)
    out = dict(zip(OP_STATS, (int(v) for v in a)))
    out.update(dict(zip(OP_STATS_EX, (int(v) for v in b[:n]))))
    return out


def _sz(seq):
    seq = [int(s) for s in seq]
    return (ctypes.c_size_t * max(len(seq), 1))(*seq)


def conv_raw(x_ptr: int, xshape, y_ptr: int, yshape, z_ptr: int, zshape, slab_lo=0, slab_hi=None, accumulate=False):
    """``gft_conv_raw`` on caller-owned device buffers (e.g. ``torch.Tensor.data_ptr()``)."""
    L = lib()
    nd = len(zshape)
    if slab_hi is None:
        slab_hi = zshape[0] if nd else 1
    rc = L.gft_conv_raw(x_ptr, _sz(xshape), y_ptr, _sz(yshape), z_ptr, _sz(zshape), nd, slab_lo, slab_hi, int(accumulate))
    if rc != 0:
        raise TaylorError((L.gft_last_error() or b"").decode())


def dist_unique_id() -> bytes:
    """Rank 0: the 128-byte RCCL unique id to hand to every rank's :func:`dist_init`."""
    buf = ctypes.create_string_buffer(128)
    if lib().gft_dist_unique_id(buf) != 0:
        raise TaylorError((lib().gft_last_error() or b"").decode())
    return buf.raw


def dist_init(rank: int, world: int, unique_id: bytes) -> None:
    """Join the library's own RCCL communicator (one process per GPU; call after :func:`init`)."""
    if lib().gft_dist_init(rank, world, ctypes.create_string_buffer(unique_id, 128)) != 0:
        raise TaylorError((lib().gft_last_error() or b"").decode())


def conv_raw_sharded(x_ptr: int, xshape, y_ptr: int, yshape, z_ptr: int, zshape):
    """``gft_conv_raw_sharded``: the product with its leading output axis sharded over the library's communicator."""
    L = lib()
    if L.gft_conv_raw_sharded(x_ptr, _sz(xshape), y_ptr, _sz(yshape), z_ptr, _sz(zshape), len(zshape)) != 0:
        raise TaylorError((L.gft_last_error() or b"").decode())


def conv_macs(xshape, yshape, zshape, slab_lo=0, slab_hi=None) -> float:
    nd = len(zshape)
    if slab_hi is None:
        slab_hi = zshape[0] if nd else 1
    return float(lib().gft_conv_macs(_sz(xshape), _sz(yshape), _sz(zshape), nd, slab_lo, slab_hi))


def plan_slabs(n0: int, world: int, rank: int):
    """Folded leading-axis slab assignment: ((lo0, hi0), (lo1, hi1), balanced_for_all_gather)."""
    out = (ctypes.c_size_t * 4)()
    even = lib().gft_plan_slabs(n0, world, rank, out)
    return (int(out[0]), int(out[1])), (int(out[2]), int(out[3])), bool(even)
