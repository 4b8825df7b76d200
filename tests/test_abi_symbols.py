"""CPU-side checks of the drop-in boundary: the C-ABI library loads, and exports every symbol
that include/gftaylor.h declares (no compute calls — there is no GPU here).  Also exercises the
pure-integer entry points (slab planner, MAC counter), which need no device."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def product_lib():
    import genfer_amd

    if not os.path.exists(genfer_amd.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    return genfer_amd.lib()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gftaylor.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(gfti?_[a-z0-9_]+)\s*\(", text)
    return sorted(set(n for n in names if n not in ("gft_poly",)))


def test_header_declares_expected_surface():
    syms = declared_symbols()
    assert len(syms) > 95
    for must in ("gft_mul", "gft_subst_var", "gft_shift_down", "gft_extend_to_dim", "gft_conv_raw", "gfti_mul"):
        assert must in syms


def test_library_exports_every_declared_symbol(product_lib):
    missing = [s for s in declared_symbols() if not hasattr(product_lib, s)]
    assert not missing, f"declared in include/gftaylor.h but not exported: {missing}"


def test_handle_api_table_matches_header():
    from genfer_amd.taylor import HANDLE_API

    syms = set(declared_symbols())
    for name in HANDLE_API:
        assert "gft_" + name in syms, name
        assert "gfti_" + name in syms, name


def test_widths(product_lib):
    assert product_lib.gft_width() == 1
    assert product_lib.gfti_width() == 2


def test_conv_macs_matches_survey_numbers():
    import genfer_amd

    n = 128
    assert genfer_amd.conv_macs([n] * 3, [n] * 3, [n] * 3) == float((n * (n + 1) // 2) ** 3) == 562741641216.0
    n = 64
    assert genfer_amd.conv_macs([n] * 4, [n] * 4, [n] * 4) == float((n * (n + 1) // 2) ** 4)
    # compact operand: x has 2 slabs on axis 2
    assert genfer_amd.conv_macs([4, 4, 2], [4, 4, 4], [4, 4, 4]) == 10 * 10 * (1 + 2 + 2 + 2)


def test_plan_slabs_partitions_and_balances():
    import genfer_amd

    for n0, world in [(64, 8), (128, 8), (128, 2), (100, 8), (7, 4), (5, 8), (1, 2)]:
        owned = [0] * n0
        works = []
        for r in range(world):
            (a, b), (c, d), even = genfer_amd.plan_slabs(n0, world, r)
            assert 0 <= a <= b <= n0 and 0 <= c <= d <= n0
            for k in list(range(a, b)) + list(range(c, d)):
                owned[k] += 1
            works.append(sum(k + 1 for k in list(range(a, b)) + list(range(c, d))))
        assert owned == [1] * n0, (n0, world, owned)
        if n0 % (2 * world) == 0:
            assert max(works) == min(works), (n0, world, works)  # folded assignment is exactly balanced


def test_integration_md_extern_block_is_generated_from_the_header():
    """INTEGRATION.md's Rust `extern "C"` block lists every declared entry point (it is generated from the header)."""
    import subprocess
    import sys

    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_rust_extern.py"), "--check"])
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for sym in declared_symbols():
        assert f"pub fn {sym}(" in doc, sym
