#!/usr/bin/env python3
"""Where the host's two threads spend a program's wall clock (GFT_TRACE_LQ=1): the launch worker inside hipLaunchKernel vs
waiting for the interpreter, the interpreter waiting for the worker.  Usage: trace_lq.py <program substring> ["--bounds"] [runs]"""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GFT_TRACE_LQ"] = "1"
import genfer_amd  # noqa: E402

only = sys.argv[1]
extra = sys.argv[2] if len(sys.argv) > 2 else ""
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
f = [p for p in sorted(glob.glob(os.path.join(ROOT, "tests/golden/sgcl/neurips2023/**/*.sgcl"), recursive=True)) if only in p][0]
src = open(f).read()
genfer_amd.init(0)
prefix = "gfti_" if "--bounds" in extra else "gft_"
for i in range(runs):
    rc, text, t = genfer_amd.run_sgcl_with_backend(src, ("--limit 100 " + extra).strip(), genfer_amd.LIB_PATH, prefix)
    assert rc == 0, text
    sys.stderr.flush()
    print(f"run {i}: Total inference time {t['time_infer']:.6f} s", file=sys.stderr, flush=True)
    genfer_amd.lib().gft_set_option(b"trace_lq_report", 1.0)
