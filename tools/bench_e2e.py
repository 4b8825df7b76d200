#!/usr/bin/env python3
"""End-to-end seconds on the NeurIPS'23 benchmark programs (BASELINE configs[2]): the host interpreter
with the HIP backend (libgftaylor) vs the same interpreter over the CPU oracle, on this box, best of N
("Total inference time" protocol of the reference's benchmarks/neurips2023/exact/bench.py:33-35,94-105).
Usage: bench_e2e.py [--limit 100] [--runs 3] [--only substr] [--gpu-only] [--bounds]"""
import ctypes
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import genfer_amd  # noqa: E402

args = sys.argv[1:]


def opt(name, default):
    for i, a in enumerate(args):
        if a == name:
            return args[i + 1]
    return default


limit = opt("--limit", "100")
runs = int(opt("--runs", "3"))
only = opt("--only", "")
gpu_only = "--gpu-only" in args
bounds = "--bounds" in args  # interval tensors (gfti_/orci_ entry points)
files = sorted(glob.glob(os.path.join(ROOT, "tests/golden/sgcl/neurips2023/**/*.sgcl"), recursive=True))
files = [f for f in files if only in f]
oracle = os.path.join(ROOT, "oracle", "liborc.so")
if not os.path.exists(oracle):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
genfer_amd.init(0)
rows = []
for f in files:
    src = open(f).read()
    first = src.splitlines()[0] if src else ""
    file_flags = first[len("# flags:"):].strip() if first.startswith("# flags:") else ""
    flags = file_flags if "--limit" in file_flags or "--no-probs" in file_flags else (file_flags + f" --limit {limit}").strip()
    row = {"program": os.path.relpath(f, os.path.join(ROOT, "tests/golden/sgcl/neurips2023")), "flags": flags}
    if bounds:
        flags += " --bounds"
    backends = (("gpu", genfer_amd.LIB_PATH, "gfti_" if bounds else "gft_"), ("cpu_oracle", oracle, "orci_" if bounds else "orc_"))
    for name, lib, prefix in backends[: 1 if gpu_only else 2]:
        best = None
        for _ in range(runs):
            before = genfer_amd.op_stats() if name == "gpu" else None
            rc, text, t = genfer_amd.run_sgcl_with_backend(src, flags, lib, prefix)
            if rc != 0:
                best = None
                row[name + "_error"] = text[-200:]
                break
            best = t["time_infer"] if best is None else min(best, t["time_infer"])
            if before is not None:  # counters of ONE run of the program
                after = genfer_amd.op_stats()
                row["gpu_op_stats_per_run"] = {k: after[k] - before[k] for k in after}
        row[name + "_s"] = best
    rows.append(row)
    print(f"{row['program']:55s} gpu {row['gpu_s']!s:>10}  cpu {row.get('cpu_oracle_s')!s:>10}  launches {row.get('gpu_op_stats_per_run', {}).get('launches')}  graph {({k: row.get('gpu_op_stats_per_run', {}).get(k) for k in ('graph_executions', 'graph_recordings', 'batch_launches', 'batch_items', 'graph_us', 'linear_scans', 'scalar_readbacks', 'chains_materialised')})}", flush=True)
print(json.dumps({"limit": limit, "runs": runs, "host_cores": os.cpu_count(), "rows": rows}))
