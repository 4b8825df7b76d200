#!/usr/bin/env python3
"""Times ONE program on the HIP backend (best of N "Total inference time") and prints the library's counters of one
run — the vehicle for `rocprofv3 --kernel-trace --stats -- python3 tools/run_sgcl.py <file.sgcl> "<flags>" [runs]`.
Paths are relative to tests/golden/sgcl/ unless they exist as given."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import genfer_amd  # noqa: E402

if os.environ.get("GENFER_HOST_LIB"):  # A/B of interpreter builds: another libgfhost
    genfer_amd.HOST_LIB_PATH = os.path.abspath(os.environ["GENFER_HOST_LIB"])
path = sys.argv[1]
flags = sys.argv[2] if len(sys.argv) > 2 else ""
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
if not os.path.exists(path):
    path = os.path.join(ROOT, "tests", "golden", "sgcl", path)
src = open(path).read()
first = src.splitlines()[0] if src else ""
if first.startswith("# flags:"):
    flags = (first[len("# flags:"):].strip() + " " + flags).strip()
genfer_amd.init(0)
prefix = "gfti_" if "--bounds" in flags.split() else "gft_"
best, stats = None, None
for _ in range(runs):
    before = genfer_amd.op_stats()
    rc, text, t = genfer_amd.run_sgcl_with_backend(src, "--no-timing " + flags, genfer_amd.LIB_PATH, prefix)
    assert rc == 0, text
    after = genfer_amd.op_stats()
    stats = {k: after[k] - before[k] for k in after}
    best = t["time_infer"] if best is None else min(best, t["time_infer"])
print(json.dumps({"program": os.path.relpath(path, ROOT), "flags": flags, "runs": runs, "gpu_s": best, "op_stats_per_run": stats}))
