"""Shared fixtures.

Backends:
  * ``oracle`` — CPU restatement of the reference (oracle/liborc.so, built on demand with
    ``make -C oracle``); always available, used as the checker.
  * ``hip``    — the product: genfer_amd/csrc/libgftaylor.so through the C ABI on a real
    MI355X.  Every test that touches it carries ``@pytest.mark.gpu``.
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_oracle_lib():
    so = os.path.join(ROOT, "oracle", "liborc.so")
    src = [os.path.join(ROOT, "oracle", f) for f in ("orc_capi.cpp", "taylor_oracle.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liborc.so"])
    return ctypes.CDLL(so)


@pytest.fixture(scope="session")
def oracle_lib():
    return load_oracle_lib()


@pytest.fixture(scope="session")
def OTP(oracle_lib):
    """Oracle-backed TaylorPoly<F64>."""
    from genfer_amd.taylor import bind

    return bind(oracle_lib, "orc_")


@pytest.fixture(scope="session")
def OTPI(oracle_lib):
    """Oracle-backed TaylorPoly<Interval<F64>>."""
    from genfer_amd.taylor import bind

    return bind(oracle_lib, "orci_")


@pytest.fixture(scope="session")
def GTP():
    """Product TaylorPoly<F64> on the GPU (fails loudly if the HIP library is missing)."""
    import genfer_amd

    return genfer_amd.TaylorPoly


@pytest.fixture(scope="session")
def GTPI():
    import genfer_amd

    return genfer_amd.IntervalTaylorPoly


BACKENDS = [pytest.param("oracle", id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


# ---- size-threshold dispatch in the tests --------------------------------------------
# The library keeps small host-built tensors on its host tier.  The GPU tests exist to check the HIP kernels on
# small shapes too, so by default every `gpu` test runs with the dispatch OFF (everything on the device).  The tests
# of TIERED_MODULES that go through the handle API run twice: [device] = kernels only, [host] = the default dispatch
# (host tier below the crossover) — same assertions, same oracle.
TIERED_MODULES = {"test_fuzz_gpu", "test_parity_gpu", "test_reference_unit_vectors", "test_exact_kats", "test_e2e_snapshots",
                  "test_horner_shapes_gpu", "test_interval_pins"}
DEVICE_ONLY = ("conv_raw", "conv_tiled", "conv_staged", "full_size", "reference_order_kernels", "on_the_tiled", "oracle_byte_exact",
               "mid_size", "test_oracle_", "limit100")


def _set_tier(name):
    import genfer_amd

    L = genfer_amd.lib()
    # GFT_TEST_HOST_LIMITS="elems,macs": override the [host] tier's thresholds (debugging aid: lift them to run whole
    # suites on the host tier)
    elems, macs = (float(t) for t in os.environ.get("GFT_TEST_HOST_LIMITS", "-1,-1").split(","))
    assert L.gft_set_option(b"host_max_elems", 0.0 if name == "device" else elems) == 0
    assert L.gft_set_option(b"host_max_macs", macs) == 0


@pytest.fixture(autouse=True)
def tier(request):
    name = getattr(request, "param", "device")
    if request.node.get_closest_marker("gpu") is not None:
        _set_tier(name)
    yield name


def pytest_generate_tests(metafunc):
    mod = metafunc.module.__name__.split(".")[-1]
    if mod in TIERED_MODULES and not any(p in metafunc.function.__name__ for p in DEVICE_ONLY):
        metafunc.parametrize("tier", ["device", pytest.param("host", marks=pytest.mark.gpu)], indirect=True)


@pytest.fixture(params=BACKENDS)
def backend(request):
    return request.param


@pytest.fixture
def TP(backend, request):
    """TaylorPoly<F64> class of the parametrised backend."""
    return request.getfixturevalue("OTP" if backend == "oracle" else "GTP")


@pytest.fixture(scope="session")
def unit_vectors():
    with open(os.path.join(GOLDEN, "unit_vectors.json")) as f:
        return json.load(f)


# ---- comparison helpers -------------------------------------------------------------

REL_TOL = 1e-10  # north_star: GPU vs reference CPU f64 path within 1e-10 relative


def assert_poly_equal(got, want, backend, rel=REL_TOL, exact_on_hip=False, scale=None):
    """`want` is a TaylorPoly of the same backend or (array, degrees) pair.

    oracle: bit-exact (the reference asserts assert_eq! on f64).
    hip: integer bookkeeping (degrees, stored shape) bit-exact; values within `rel` relative
    (normwise against `scale` when given), or bit-exact when `exact_on_hip`.
    """
    if isinstance(want, tuple):
        want_arr, want_deg = np.asarray(want[0], dtype=np.float64), tuple(want[1])
    else:
        want_arr, want_deg = want.array(), want.degrees_p1()
    got_arr = got.array()
    assert got.degrees_p1() == want_deg, f"degrees_p1 {got.degrees_p1()} != {want_deg}"
    assert got_arr.shape == want_arr.shape, f"stored shape {got_arr.shape} != {want_arr.shape}"
    if backend == "oracle" or exact_on_hip:
        if not np.array_equal(got_arr, want_arr):
            raise AssertionError(f"not bit-exact:\n got={got_arr!r}\nwant={want_arr!r}")
    else:
        ref = np.abs(want_arr) if scale is None else np.asarray(scale)
        err = np.abs(got_arr - want_arr)
        bound = rel * np.maximum(ref, np.finfo(np.float64).tiny)
        ok = (err <= bound) | (got_arr == want_arr)
        if not ok.all():
            i = np.unravel_index(np.argmax(err / np.maximum(bound, 1e-300)), err.shape)
            raise AssertionError(f"rel error too large at {i}: got {got_arr[i]!r} want {want_arr[i]!r}")


def splitmix64_uniform(seed: int, n: int) -> np.ndarray:
    """SURVEY §8(d) synthetic inputs: splitmix64 -> (u >> 11) * 2^-53 in [0,1), row-major.

    splitmix64's state after i+1 steps is seed + (i+1)*0x9E3779B97F4A7C15 (mod 2^64), so the
    stream vectorises; uint64 arithmetic wraps exactly like the C reference generator.
    """
    with np.errstate(over="ignore"):
        x = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + np.uint64(0x9E3779B97F4A7C15) * np.arange(1, n + 1, dtype=np.uint64)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
