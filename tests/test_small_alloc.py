"""gft_small_alloc.hpp (the host side's small-block free lists) under AddressSanitizer + UBSan + leak check on the CPU."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_small_block_lists_under_asan():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "small_alloc_check")
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-pthread",
                               os.path.join(ROOT, "tests", "small_alloc_check.cpp"), "-o", exe])
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
        assert r.returncode == 0, r.stdout + r.stderr
        assert "small_alloc ok" in r.stdout
