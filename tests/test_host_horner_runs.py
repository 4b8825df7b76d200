"""CPU check of the host tier's Horner step in runs (gft_host.hpp, round 6): tests/host_horner_check.hip compares it bit for bit
with the element-by-element form on thousands of random cases.  Built with hipcc, run on the host — no GPU needed."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_horner_runs_equal_the_element_form(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "host_horner_check")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-std=c++17", "-O2", "-ffp-contract=off", "-Wno-unused-function", "-o", exe,
                           os.path.join(ROOT, "tests", "host_horner_check.hip")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "host_horner ok" in out.stdout, out.stdout + out.stderr
