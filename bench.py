#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path: TaylorPoly mul f64 (dense truncated N-d
polynomial product, src/multivariate_taylor.rs:971-1072) in GMAC/s on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c4|small]

A "step" is one full product z = x (*) y of the workload with x, y already resident in HBM
(synthetic splitmix64 inputs).  N > 1: one process per GPU (torchrun / RANK env);
the product's leading output axis is sharded with the folded slab assignment
(gft_plan_slabs), operands replicated, result slabs exchanged with RCCL all-gather inside the
timed region => "strong" scaling of one product.

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (the product kernel
against the gfx950 FP64 FMA peak — the product is compute-bound, SURVEY §8d — with the
algorithmic-HBM figure beside it) and `cpu_baseline` (the oracle = scalar restatement of the
reference loop nest, timed on this box's host cores on a bounded sample; N = 1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # before anything initialises HIP (see genfer_amd/__init__.py)

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6  # MI355X vendor FP64 peak, vector == matrix (256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

WORKLOADS = {
    # name: (shape, description)  — c2 is BASELINE.json configs[1], the config the metric is quoted on
    "c2": ([128, 128, 128], "synthetic 3-var TaylorPoly mul, order 128 each (128^3 f64 coeffs)"),
    "c4": ([64, 64, 64, 64], "4-var order-64 mul (64^4 coeffs)"),
    "small": ([32, 32, 32], "3-var order-32 mul (debug size)"),
}


def splitmix64_uniform(seed, n):
    with np.errstate(over="ignore"):
        x = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * np.arange(1, n + 1, dtype=np.uint64)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def cpu_baseline(shape, x, y, budget_hint_s=20.0):
    """Time the oracle (oracle/liborc.so: same loop nest / summation order as mt:971-1012, one
    thread, -O2 -ffp-contract=off) on a bounded sample of the SAME product: a few leading-axis
    output slabs, chosen so that the sample is ~10-30 s of CPU work."""
    so = os.path.join(ROOT, "oracle", "liborc.so")  # built by ensure_oracle() before the GPU was initialised
    lib = ctypes.CDLL(so)
    lib.orc_mul_slabs_timed.restype = ctypes.c_double
    szp = ctypes.POINTER(ctypes.c_size_t)
    lib.orc_mul_slabs_timed.argtypes = [ctypes.c_void_p, szp, ctypes.c_void_p, szp, ctypes.c_void_p, szp,
                                        ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t,
                                        ctypes.POINTER(ctypes.c_double)]
    nd = len(shape)
    sz = (ctypes.c_size_t * nd)(*shape)
    res = np.zeros(shape)
    macs = ctypes.c_double(0.0)
    n0 = shape[0]
    # calibrate on the cheapest slab, then pick slabs {mid, top} if they fit the budget
    t = lib.orc_mul_slabs_timed(x.ctypes.data_as(ctypes.c_void_p), sz, y.ctypes.data_as(ctypes.c_void_p), sz,
                                res.ctypes.data_as(ctypes.c_void_p), sz, nd, 0, 1, ctypes.byref(macs))
    rate = macs.value / max(t, 1e-9)
    per_slab_unit = macs.value  # MACs of slab 0; slab k costs (k+1)x
    slabs = []
    spent = 0.0
    for k in (n0 - 1, n0 // 2 - 1, n0 // 4 - 1):
        if k < 0:
            continue
        cost = per_slab_unit * (k + 1) / rate
        if spent + cost <= budget_hint_s or not slabs:
            slabs.append(k)
            spent += cost
    total_macs, total_t = 0.0, 0.0
    for k in slabs:
        res[k] = 0.0
        t = lib.orc_mul_slabs_timed(x.ctypes.data_as(ctypes.c_void_p), sz, y.ctypes.data_as(ctypes.c_void_p), sz,
                                    res.ctypes.data_as(ctypes.c_void_p), sz, nd, k, k + 1, ctypes.byref(macs))
        total_macs += macs.value
        total_t += t
    return {
        "value": total_macs / total_t / 1e9,
        "unit": "GMAC/s",
        "cores": 1,
        "kind": "port",
        "sample": f"leading-axis output slabs k0 in {sorted(slabs)} of the same product "
                  f"({total_macs:.3e} of {per_slab_unit * n0 * (n0 + 1) / 2:.3e} MACs, {total_t:.1f} s, "
                  f"host has {os.cpu_count()} cores; reference is single-threaded, src/main.rs:101-105)",
    }, res, slabs


def cpu_baseline_all_cores(shape, x, y, max_threads=64):
    """The stronger CPU baseline of SURVEY §8d: the same oracle loop nest, different leading-axis output slabs on
    different host threads (slabs are independent; per-element operation order unchanged).  One of the heaviest
    slabs per thread, so the sample is ~4 s of wall time.  ctypes releases the GIL during the call."""
    from concurrent.futures import ThreadPoolExecutor

    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "liborc.so"))
    lib.orc_mul_slabs_timed.restype = ctypes.c_double
    szp = ctypes.POINTER(ctypes.c_size_t)
    lib.orc_mul_slabs_timed.argtypes = [ctypes.c_void_p, szp, ctypes.c_void_p, szp, ctypes.c_void_p, szp,
                                        ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t,
                                        ctypes.POINTER(ctypes.c_double)]
    nd, n0 = len(shape), shape[0]
    threads = max(1, min(max_threads, os.cpu_count() or 1, n0))
    sz = (ctypes.c_size_t * nd)(*shape)
    res = np.zeros(shape)
    slabs = list(range(n0 - threads, n0))

    def one(k):
        m = ctypes.c_double(0.0)
        lib.orc_mul_slabs_timed(x.ctypes.data_as(ctypes.c_void_p), sz, y.ctypes.data_as(ctypes.c_void_p), sz,
                                res.ctypes.data_as(ctypes.c_void_p), sz, nd, k, k + 1, ctypes.byref(m))
        return m.value

    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        macs = sum(ex.map(one, slabs))
    wall = time.perf_counter() - t0
    return {
        "value": macs / wall / 1e9,
        "unit": "GMAC/s",
        "cores": threads,
        "kind": "port",
        "sample": f"leading-axis output slabs k0 in [{slabs[0]}, {slabs[-1]}] of the same product, one per thread "
                  f"({macs:.3e} MACs, {wall:.1f} s wall, host has {os.cpu_count()} cores)",
    }


PROFILE_ROUNDS = ("r06", "r05", "r04", "r03", "r02", "r01")  # newest first


def pmc_traffic(world, workload):
    """HBM-side bytes per product launch from the committed rocprofv3 PMC passes of this same command AND workload
    (profiles/rNN/pmc_k_conv_tiled_<workload>.json: FETCH_SIZE and WRITE_SIZE in KB, separate passes; FETCH_SIZE
    doubled per MI355X_MICROARCH.md §HBM — gfx950 reports half of wide coalesced reads).  PMC counters cannot be
    read from inside this process; null when no profile of this workload is committed (and for N > 1)."""
    if world != 1:
        return None, "not collected for N > 1"
    for rnd in PROFILE_ROUNDS:
        names = [f"pmc_k_conv_tiled_{workload}.json"] + (["pmc_k_conv_tiled.json"] if workload == "c2" else [])
        for name in names:
            path = os.path.join(ROOT, "profiles", rnd, name)
            if not os.path.exists(path):
                continue
            d = json.load(open(path))
            if d.get("workload", "c2") != workload:
                continue
            try:
                return ((2.0 * d["FETCH_SIZE"]["per_launch_mean"] + d["WRITE_SIZE"]["per_launch_mean"]) * 1024.0,
                        f"committed rocprofv3 --pmc passes of this command, profiles/{rnd}/{name} (2 x FETCH_SIZE + WRITE_SIZE per "
                        "launch); NOT measured by this run — PMC counters cannot be read from inside the process")
            except KeyError:
                continue
    return None, "no committed PMC profile of this workload"


# (name, program under tests/golden/sgcl/, flags, CPU-oracle runs, committed oracle report or None, recorded CPU seconds or
# None).  The first four are the NeurIPS'23 programs BASELINE names; two_populations2000 is the reference's own slow/ fixture of
# general Horner loops; three_ / four_populations are this repo's programs in which rank-3 / rank-4 GENERAL products dominate.
# Every row is CHECKED at the size it is timed at: against the oracle report computed in the same call, or — where the oracle
# needs minutes — against the committed oracle report (tests/golden/make_c3_limit100_golden.py).
E2E_PROGRAMS = (
    ("hmm", "neurips2023/approx/hmm/hmm.sgcl", "--limit 100", 2, None, None),
    ("mixture", "neurips2023/approx/mixture/mixture.sgcl", "--limit 100", 1, None, None),
    ("two_populations", "neurips2023/approx/two_populations/two_populations.sgcl", "--limit 100", 2, None, None),
    ("switchpoint", "neurips2023/approx/switchpoint/switchpoint.sgcl", "--limit 100", 2, None, None),
    ("two_populations2000", "test_expect/slow/two_populations2000.sgcl", "", 2, None, None),
    ("three_populations", "bench/three_populations.sgcl", "--limit 100", 1, "three_populations.oracle.txt", None),
    ("four_populations", "bench/four_populations.sgcl", "--limit 24", 1, "four_populations.oracle.txt", None),
    # TaylorPoly<Interval<F64>> (`--bounds`): the CPU oracle needs ~26 s for hmm (run once here) and 12 minutes for mixture
    # (recorded figure + committed oracle report)
    ("hmm_bounds", "neurips2023/approx/hmm/hmm.sgcl", "--limit 100 --bounds", 1, "hmm-bounds.oracle.txt", None),
    ("mixture_bounds", "neurips2023/approx/mixture/mixture.sgcl", "--limit 100 --bounds", 0, "mixture-bounds.oracle.txt",
     (717.0, "profiles/r02/e2e_neurips_limit100_bounds_with_cpu_oracle_earlier_collection.json (round 2, another box)")),
)


def e2e_seconds(gpu_runs=5, oracle_available=True):
    """BASELINE's second metric: end-to-end seconds ("Total inference time", best of N — the protocol of the
    reference's benchmarks/neurips2023/exact/bench.py:33-35,94-105) on NeurIPS'23 programs at --limit 100: the host
    interpreter over libgftaylor (`backend_s`, best of 5; `host_tier_frac` says how much of it ran on the library's host tier) and, beside it, the same interpreter over the CPU oracle on this
    box's host (1 thread; best of 2, a single run for the long ones).  `parity`: the GPU report of the timed
    configuration against the oracle's report (genfer_amd/reports.py: 1e-10 on primary quantities) — "ok" or the first
    difference; a difference makes bench.py exit non-zero."""
    import genfer_amd
    from genfer_amd.reports import first_difference

    oracle = os.path.join(ROOT, "oracle", "liborc.so")
    rows = {}
    failed = False
    for name, rel, flags, cpu_runs, stored, recorded in E2E_PROGRAMS:
        src = open(os.path.join(ROOT, "tests", "golden", "sgcl", rel)).read()
        first = src.splitlines()[0] if src else ""
        if first.startswith("# flags:"):  # the fixture's own flags (tests/integration.rs protocol)
            flags = (first[len("# flags:"):].strip() + " " + flags).strip()
        row = {"flags": flags}
        run_flags = "--no-timing " + flags  # the seconds come back beside the report; the report then has no clock in it
        texts = {}
        if not oracle_available:
            cpu_runs = 0
        for key, lib, prefix, runs in (("backend_s", genfer_amd.LIB_PATH, "gft_", gpu_runs if cpu_runs else min(gpu_runs, 3)),
                                       ("cpu_oracle_s", oracle, "orc_", cpu_runs)):
            best = None
            for _ in range(runs):
                before = genfer_amd.op_stats() if key == "backend_s" else None
                pfx = prefix[:-1] + "i_" if "--bounds" in flags.split() else prefix  # the Interval<F64> entry points
                rc, text, t = genfer_amd.run_sgcl_with_backend(src, run_flags, lib, pfx)
                if rc != 0:
                    row[key + "_error"] = text[-200:]
                    best = None
                    if key == "backend_s":
                        failed = True
                    break
                texts[key] = text
                best = t["time_infer"] if best is None else min(best, t["time_infer"])
                if before is not None:  # what one run of the program costs (the same every run)
                    after = genfer_amd.op_stats()
                    for k in ("launches", "host_tier_ops", "deferred_ops", "tiled", "staged", "per_output", "linear_scans", "fused_observe_adds", "nested_adds", "scans_proven", "graph_executions", "batch_launches", "batch_items"):
                        row[k] = after[k] - before[k]
            row[key] = best
            row[key.replace("_s", "_runs")] = runs
            if key == "backend_s" and "launches" in row:
                # where the TaylorPoly operations of this program ran: a row that is (almost) all host tier — switchpoint, the
                # exact/ programs — is a CPU number of the library's host tier, not a GPU result
                row["host_tier_frac"] = round(row["host_tier_ops"] / max(1, row["host_tier_ops"] + row["launches"]), 4)
        if recorded and row.get("cpu_oracle_s") is None:
            row["cpu_oracle_s_recorded"] = {"value": recorded[0], "source": recorded[1]}
        # parity of the timed configuration, at the timed size
        want, against = None, None
        if "cpu_oracle_s" in texts:
            want, against = texts["cpu_oracle_s"], "the oracle's report computed in this call"
        elif stored and os.path.exists(os.path.join(ROOT, "tests", "golden", "c3_limit100", stored)):
            want = open(os.path.join(ROOT, "tests", "golden", "c3_limit100", stored)).read()
            against = f"committed oracle report tests/golden/c3_limit100/{stored}"
        if "backend_s" in texts and want is not None:
            diff = first_difference(texts["backend_s"], want)
            row["parity"] = "ok" if diff is None else diff
            row["parity_against"] = against
            if diff is not None:
                failed = True
        else:
            row["parity"] = "unchecked (no oracle report available)"
        rows[name] = row
    return {"unit": "s", "protocol": "best-of-N Total inference time (flags per program)", "programs": rows,
            "parity_failed": failed}


_CLOCK_HELPER = r"""
import glob, json, os, select, shutil, subprocess, sys, time
# Two services for a parent that must not fork after it has initialised the GPU:
#   "start" ... "stop": poll the amdgpu sysfs nodes (shader clock, socket power) every ~10 ms while the parent's TIMED
#                       loop runs -> one JSON line of samples per card (no exec, no GPU access: plain file reads);
#   "go":               one `rocm-smi --showclocks --showpower` (the round-3 sampler, run under a separate untimed load).
def cards():
    out = []
    for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        try:
            if open(dev + "/vendor").read().strip() != "0x1002":
                continue
        except OSError:
            continue
        hw = sorted(glob.glob(dev + "/hwmon/hwmon*"))
        f = [h + "/freq1_input" for h in hw if glob.glob(h + "/freq1_input")]
        p = [h + "/" + n for h in hw for n in ("power1_average", "power1_input") if glob.glob(h + "/" + n)]
        out.append({"dev": dev, "freq": f[0] if f else None, "power": p[0] if p else None, "dpm": dev + "/pp_dpm_sclk"})
    return out
def read_num(path):
    try:
        return float(open(path).read().split()[0])
    except Exception:
        return None
def read_dpm(path):
    try:
        for ln in open(path).read().splitlines():
            if ln.rstrip().endswith("*"):
                return float("".join(ch for ch in ln.split(":")[1] if ch.isdigit() or ch == "."))
    except Exception:
        pass
    return None
exe = shutil.which("rocm-smi")
cs = cards()
# (commands are read UNBUFFERED from fd 0: select() below looks at the descriptor, and a "stop" that arrived in the same
# pipe read as "start" would otherwise sit in Python's buffer where select() never sees it)
pending = b""
def next_cmd(timeout=None):
    global pending
    while b"\n" not in pending:
        if timeout is not None:
            r, _, _ = select.select([0], [], [], timeout)
            if not r:
                return None
        chunk = os.read(0, 4096)
        if not chunk:
            return ""
        pending += chunk
    line, _, pending = pending.partition(b"\n")
    return line.decode().strip() or " "
while True:
    cmd = next_cmd()
    if cmd == "":
        break
    if cmd == "start":
        samples = [{"mhz": [], "w": []} for _ in cs]
        t0 = time.time()
        while True:
            for c, s in zip(cs, samples):
                hz = read_num(c["freq"]) if c["freq"] else None
                mhz = hz / 1e6 if hz else read_dpm(c["dpm"])
                uw = read_num(c["power"]) if c["power"] else None
                if mhz is not None: s["mhz"].append(mhz)
                if uw is not None: s["w"].append(uw / 1e6)
            if next_cmd(0.01) is not None:
                break
            if time.time() - t0 > 120:
                break
        sys.stdout.write(json.dumps({"seconds": time.time() - t0, "cards": samples}) + "\n")
        sys.stdout.flush()
    elif cmd == "go":
        if exe:
            try:
                sys.stdout.write(subprocess.run([exe, "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout)
            except Exception:
                pass
        break
"""


def timed_loop_samples(helper, text):
    """The sysfs samples the helper took between "start" and "stop" (i.e. DURING the timed loop), reduced to the card that
    drew the most power (the one under load when the box exposes several)."""
    try:
        rec = json.loads(text)
        best = None
        for c in rec["cards"]:
            if not c["mhz"] and not c["w"]:
                continue
            key = float(np.mean(c["w"])) if c["w"] else 0.0
            if best is None or key > best[0]:
                best = (key, c)
        if best is None:
            return None
        c = best[1]
        out = {"samples": max(len(c["mhz"]), len(c["w"])), "window_s": rec["seconds"], "source": "amdgpu sysfs (hwmon freq1_input / power1_average), polled by a child forked before GPU initialisation"}
        if c["mhz"]:
            # (the loop starts on a part that was idling: the first samples are the ramp — `steady` is the median of the second
            # half of the window, what the kernel runs at once the clock has settled)
            out["sclk_mhz"] = {"min": float(np.min(c["mhz"])), "median": float(np.median(c["mhz"])), "max": float(np.max(c["mhz"])),
                               "steady": float(np.median(c["mhz"][len(c["mhz"]) // 2:]))}
        if c["w"]:
            out["socket_power_w"] = {"min": float(np.min(c["w"])), "median": float(np.median(c["w"])), "max": float(np.max(c["w"]))}
        return out
    except Exception:  # noqa: BLE001 - informational only
        return None


def ensure_oracle():
    """The checker / CPU baseline (oracle/liborc.so) must exist BEFORE this process touches the GPU: building it spawns
    `make`, and a process that has initialised HIP must not fork + exec on this pool.  *.so files are git-ignored, so a
    fresh checkout needs this.  Returns whether the library is there; without it the CPU legs are skipped with a note."""
    import subprocess

    so = os.path.join(ROOT, "oracle", "liborc.so")
    if os.path.exists(so):
        return True
    try:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    except Exception:  # noqa: BLE001
        return False
    return os.path.exists(so)


def start_clock_helper():
    """A child that will run `rocm-smi` when told to.  It is started BEFORE this process touches the GPU: a process that has
    initialised HIP must not fork + exec (this pool refuses it, and it can take the box down), so the sampler cannot be
    spawned at the time of the sample.  Returns None where that is not possible (e.g. under rocprofv3 --pmc, whose preloaded
    tool has initialised the GPU before main() runs — use --no-clock there)."""
    import subprocess
    try:
        return subprocess.Popen([sys.executable, "-c", _CLOCK_HELPER], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                stderr=subprocess.DEVNULL, text=True)
    except Exception:  # noqa: BLE001 - informational only
        return None


def sclk_under_load(one_step, helper):
    """The shader clock and socket power while the product runs (untimed, after the timed region): `peak` is the
    contract's 78.6 TFLOP/s at the nominal 2.4 GHz; under sustained FP64 load the part runs at its power limit below that
    (profiles/r03/clock_under_load.txt), so the line also states the fraction of the peak at the clock actually observed.
    Informational: `frac` stays achieved / peak."""
    import re

    import torch
    if helper is None or helper.poll() is not None:
        return None
    try:
        for _ in range(50):  # ~1 s of load before the sample
            one_step()
        helper.stdin.write("go\n")
        helper.stdin.flush()
        t_end = time.perf_counter() + 25.0
        while helper.poll() is None and time.perf_counter() < t_end:
            for _ in range(5):
                one_step()
            torch.cuda.synchronize()
        if helper.poll() is None:
            helper.kill()
            return None
        text = helper.stdout.read()
        m = re.search(r"sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", text)
        w = re.search(r"Package Power \(W\):\s*([0-9.]+)", text)
        out = {}
        if m:
            out["sclk_mhz_under_load"] = int(m.group(1))
        if w:
            out["socket_power_w_under_load"] = float(w.group(1))
        return out or None
    except Exception:  # noqa: BLE001 - informational only
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end seconds of the NeurIPS'23 programs (N = 1 only)")
    ap.add_argument("--conv-mode", type=int, default=0, help="0 auto, 1 reference-order kernel, 2 tiled kernel")
    ap.add_argument("--no-clock", action="store_true", help="do not sample the shader clock under load (rocm-smi)")
    args = ap.parse_args()

    # (before anything initialises the GPU: see start_clock_helper / ensure_oracle — nothing below spawns a process)
    have_oracle = ensure_oracle() if int(os.environ.get("RANK", "0")) == 0 else os.path.exists(os.path.join(ROOT, "oracle", "liborc.so"))
    clock_helper = None
    if not args.no_clock and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        clock_helper = start_clock_helper()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
    device = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("GFT_BENCH_BACKEND", "nccl")  # "gloo" only to smoke-test N > 1 on a 1-GPU box
        kw = {"device_id": torch.device("cuda", device)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)

    import genfer_amd

    genfer_amd.init(device)
    L = genfer_amd.lib()
    L.gft_set_conv_mode(args.conv_mode)
    # One explicit (non-default) stream for everything: the library's kernels, torch's copies and the
    # collectives (ProcessGroupNCCL orders its work after the current stream) are all ordered on it.
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    L.gft_set_stream(ctypes.c_void_p(stream.cuda_stream))

    shape, desc = WORKLOADS[args.workload]
    n = int(np.prod(shape))
    xh = splitmix64_uniform(1 if args.workload != "c4" else 3, n).reshape(shape)
    yh = splitmix64_uniform(2 if args.workload != "c4" else 4, n).reshape(shape)
    x = torch.from_numpy(xh).cuda()
    y = torch.from_numpy(yh).cuda()
    z = torch.zeros(shape, dtype=torch.float64, device="cuda")
    total_macs = genfer_amd.conv_macs(shape, shape, shape)
    alg_bytes = 3 * n * 8  # read x, read y, write z once

    from genfer_amd.dist import gpu_conv_slabs, local_ranges, sharded_conv

    # N > 1: the sharded product runs behind the C ABI (gft_conv_raw_sharded: the library's own RCCL communicator,
    # in-place all-gather + point-to-point exchange on its stream; torch.distributed only hands the 128-byte unique id
    # to the ranks and times the job).  GFT_BENCH_EXCHANGE=torch selects the torch.distributed exchange of
    # genfer_amd/dist.py instead; it is also the fallback if the library's communicator cannot be created.
    exchange = "none" if world == 1 else os.environ.get("GFT_BENCH_EXCHANGE", "abi")
    exchange_note = None
    if exchange == "abi":
        # Every rank executes the same collectives whatever fails where: the id travels in a broadcast that always
        # happens (None on failure), and the ranks agree on the outcome of each phase with an all-reduce BEFORE any of
        # them enters ncclCommInitRank (which would hang if a peer never calls it).
        ids, err = [None], None
        if rank == 0:
            try:
                ids = [genfer_amd.dist_unique_id()]
            except Exception as e:  # noqa: BLE001
                err = f"gft_dist_unique_id failed ({e})"
        dist.broadcast_object_list(ids, src=0)
        ok = torch.tensor([0 if ids[0] is None else 1], device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            try:
                genfer_amd.dist_init(rank, world, ids[0])
            except Exception as e:  # noqa: BLE001 - a rank that fails here fails on every rank or aborts the job (RCCL)
                err = f"gft_dist_init failed ({e})"
            ok = torch.tensor([0 if err else 1], device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            exchange = "torch"
            exchange_note = (err or "another rank could not create the C-ABI communicator") + "; torch.distributed exchange used"

    # every rank proves the exchange (both sharded entries, even and uneven splits, bit for bit against its own full
    # product) before anything is timed; a mismatch aborts the job non-zero
    selftest = None
    if exchange == "abi":
        selftest = "ok" if L.gft_dist_selftest() == 0 else (L.gft_last_error() or b"failed").decode()
        if selftest != "ok":
            print(json.dumps({"error": "gft_dist_selftest failed", "rank": rank, "detail": selftest}), flush=True)
            sys.exit(2)

    g0, g1, even, launches = local_ranges(shape[0], world, rank)
    local_macs = sum(genfer_amd.conv_macs(shape, shape, shape, a, b) for a, b in (g0, g1) if b > a)

    kern_ms = []
    EV_STEPS = 21  # event slots 3i / 3i+1 / 3i+2 of timed step i: start, local product launches done, exchange done; read
    timed_steps = [0]  # AFTER the timed region so that the event queries do not put a host round trip between steps

    def step(timed):
        i = timed_steps[0]
        rec = timed and i < EV_STEPS
        if exchange != "torch":
            L.gft_set_option(b"dist_event_slot", float(3 * i) if rec else -1.0)
            genfer_amd.conv_raw_sharded(x.data_ptr(), shape, y.data_ptr(), shape, z.data_ptr(), shape)
        else:
            sharded_conv(x, y, z, gpu_conv_slabs,
                         before_local=(lambda: L.gft_event_record(3 * i)) if rec else None,
                         after_local=(lambda: L.gft_event_record(3 * i + 1)) if rec else None)
            if rec:
                L.gft_event_record(3 * i + 2)
        if timed:
            timed_steps[0] += 1

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    sampling = False
    if clock_helper is not None and clock_helper.poll() is None:
        try:  # (a pipe write: the child polls sysfs while the timed loop runs; nothing is spawned)
            clock_helper.stdin.write("start\n")
            clock_helper.stdin.flush()
            sampling = True
        except Exception:  # noqa: BLE001
            sampling = False
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    during = None
    if sampling:
        try:
            clock_helper.stdin.write("stop\n")
            clock_helper.stdin.flush()
            during = timed_loop_samples(clock_helper, clock_helper.stdout.readline())
        except Exception:  # noqa: BLE001
            during = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    n_ev = min(args.steps, EV_STEPS)
    kern_ms = [L.gft_event_elapsed_ms(3 * i, 3 * i + 1) for i in range(n_ev)]
    exch_ms = [L.gft_event_elapsed_ms(3 * i + 1, 3 * i + 2) for i in range(n_ev)]
    ms_per_step = elapsed / args.steps * 1e3
    value = total_macs * args.steps / elapsed / 1e9
    k_ms = float(np.mean(kern_ms))
    achieved_tflops = 2.0 * local_macs / (k_ms * 1e-3) / 1e12
    traffic, traffic_src = pmc_traffic(world, args.workload)
    out = {
        "metric": "TaylorPoly mul f64 GMAC/s",
        "value": value,
        "unit": "GMAC/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "hip_force_dev_kernarg": os.environ.get("HIP_FORCE_DEV_KERNARG"),  # process-wide launch setting this run had
        "data": "synthetic (splitmix64 uniform [0,1), seeds 1/2, row-major; SURVEY §8d)",
        "config": {
            "workload": f"{args.workload}: {desc}; z = x (*) y truncated at degrees_p1 = shape",
            "shape": shape,
            "macs": total_macs,
            "parallelism": "single GPU" if world == 1 else f"leading output axis folded-sharded over {world} GPUs, "
                           f"operands replicated, RCCL {'all-gather + point-to-point' if even else 'all-reduce'} of result slabs "
                           f"({'inside libgftaylor (gft_conv_raw_sharded)' if exchange == 'abi' else 'torch.distributed'})",
        },
        "roofline": {
            "bound": "valu_fma_f64",  # (compute-bound, but not on the matrix pipe: see bound_detail)
            "bound_detail": "FP64 FMA issue rate of the vector pipe (v_fma_f64); no MFMA instruction is issued — on gfx950 "
                            "the FP64 matrix peak is the same 78.6 TFLOP/s and measured lower (profiles/r02/microbench_fp64.txt)",
            "achieved": achieved_tflops,
            "peak": FP64_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved_tflops / FP64_PEAK_TFLOPS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "note": "FP64 FMA roof (vector == matrix FP64 peak on gfx950, 78.6 TFLOP/s); flops = 2*MACs of the "
                    "slabs this rank computes / mean HIP-event duration of the product launch(es) on its stream",
            "kernel_ms": k_ms,
            # the per-step HIP-event durations of the timed steps: a slow line explains itself (first steps at a cold clock,
            # a box whose clock sits lower, one outlier)
            "kernel_ms_steps": {"min": float(np.min(kern_ms)), "median": float(np.median(kern_ms)), "max": float(np.max(kern_ms)),
                                "first": float(kern_ms[0]), "n": len(kern_ms)},
            "during_timed_loop": during,
            "hbm_algorithmic": {
                "achieved": alg_bytes * (local_macs / total_macs) / (k_ms * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": alg_bytes * (local_macs / total_macs) / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "bytes": alg_bytes,
            },
        },
    }

    if rank == 0 and world == 1:
        clk = sclk_under_load(lambda: step(False), clock_helper)
        if clk:
            out["roofline"].update(clk)
            if clk.get("sclk_mhz_under_load"):
                peak_at = FP64_PEAK_TFLOPS * clk["sclk_mhz_under_load"] / 2400.0
                out["roofline"]["frac_at_observed_clock"] = achieved_tflops / peak_at
        if during and during.get("sclk_mhz"):  # the clock of the timed loop itself, when sysfs gives it
            out["roofline"]["frac_at_timed_loop_clock"] = achieved_tflops / (FP64_PEAK_TFLOPS * during["sclk_mhz"]["steady"] / 2400.0)
    if clock_helper is not None and clock_helper.poll() is None:
        try:
            clock_helper.stdin.close()  # never told to sample: let it go
        except Exception:  # noqa: BLE001
            pass
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not have_oracle:
        out["cpu_baseline"] = None
        out["cpu_baseline_note"] = "oracle/liborc.so is missing and could not be built before GPU initialisation; CPU legs skipped"
    elif rank == 0 and world == 1 and not args.no_cpu_baseline:
        base, ref_slabs, slabs = cpu_baseline(shape, xh, yh)
        out["cpu_baseline"] = base
        out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(shape, xh, yh)
        # the timed sample doubles as an end-of-run parity check of the full-size result
        zh = z.cpu().numpy()
        worst = 0.0
        for k in slabs:
            err = np.abs(zh[k] - ref_slabs[k]) / np.abs(ref_slabs[k])
            worst = max(worst, float(err.max()))
        out["parity_max_rel_err_vs_oracle_sample"] = worst
        if worst > 1e-10:
            out["parity_failed"] = True
    else:
        # N > 1: validate the exchange — every rank recomputes the whole product locally (untimed) and
        # compares it with the gathered result
        z_local = torch.empty_like(z)
        gpu_conv_slabs(x, y, z_local, 0, shape[0])
        torch.cuda.synchronize()
        err = ((z_local - z).abs() / z_local.abs().clamp_min(1e-300)).max().reshape(1)
        if world > 1:
            dist.all_reduce(err, op=dist.ReduceOp.MAX)
        if rank == 0:
            out["cpu_baseline"] = None
            if world > 1:
                # stream-K split points differ between a full launch and a slab-range launch, so the two
                # results agree to rounding (1e-10 bar), not bit for bit
                out["sharded_vs_single_gpu_max_rel_err"] = float(err.item())
                if float(err.item()) > 1e-10:
                    out["parity_failed"] = True

    if world > 1:
        # proof that RCCL saw every rank: ncclCommCount of the library's communicator on every rank (C-ABI exchange),
        # gathered to rank 0 — together with every rank's own kernel-only and exchange-only milliseconds per step (HIP
        # events on its stream), so that the scaling record explains its own curve
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"comm_count": int(L.gft_dist_comm_count()), "kernel_ms": float(np.mean(kern_ms)),
                                          "exchange_ms": float(np.mean(exch_ms)), "local_macs": local_macs,
                                          "selftest": selftest})
        counts = [r["comm_count"] for r in per_rank]
        if rank == 0:
            out["exchange"] = exchange
            out["exchange_requested"] = os.environ.get("GFT_BENCH_EXCHANGE", "abi")
            if exchange != out["exchange_requested"]:
                out["exchange_fell_back"] = True  # the library's RCCL communicator could not be created: see exchange_note
            out["per_rank"] = per_rank
            out["rccl_comm_count_per_rank"] = counts
            if exchange_note:
                out["exchange_note"] = exchange_note
    if rank == 0 and world == 1 and not args.no_e2e and args.workload == "c2":
        out["e2e"] = e2e_seconds(oracle_available=have_oracle)
        if out["e2e"].pop("parity_failed"):
            out["parity_failed"] = True
    bad = bool(out.get("parity_failed")) or any(t < 0 for t in kern_ms)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    if bad:  # a wrong product (or a failed event query) must not look like a successful run
        sys.exit(1)


if __name__ == "__main__":
    main()
