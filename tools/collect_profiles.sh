#!/bin/bash
# Regenerates the rocprofv3 evidence bench.py's roofline block refers to.  Run on the GPU box from the repo root:
#   bash tools/collect_profiles.sh            (writes under gpurun_out/prof/, summaries as *.csv / *.json)
# Kernel trace and every PMC group are SEPARATE runs (counters perturb timing; the pool forbids mixing --pmc with
# the API trace domains).  Copy gpurun_out/prof/summary/* into profiles/rNN/ afterwards.
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT/summary"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gft_kt -o kt -- python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-clock > "$OUT/kt.log" 2>&1
cp "$(find /tmp/gft_kt -name '*kernel_stats.csv' | head -1)" "$OUT/summary/kernel_stats.csv"
grep -o '{"metric".*' "$OUT/kt.log" | tail -1 > "$OUT/summary/bench_c2_n1_under_rocprof.json"
n=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD"; do
  n=$((n+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d /tmp/gft_pmc_$n -o pmc -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-clock > "$OUT/pmc_$n.log" 2>&1
  f=$(find /tmp/gft_pmc_$n -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/pmc_$n.csv"
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
agg = {}
for f in sorted(glob.glob(out + "/pmc_*.csv")):
    per = {}
    for r in csv.DictReader(open(f)):
        if "k_conv_tiled" not in r["Kernel_Name"]:
            continue
        per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for c, d in per.items():
        v = list(d.values())
        agg[c] = {"per_launch_mean": sum(v) / len(v), "launches": len(v)}
agg["workload"] = "c2"
agg["_note"] = ("rocprofv3 --pmc passes (separate runs) of `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e` on MI355X; "
                "per-launch sums over all XCDs/SEs for the k_conv_tiled dispatches (128^3 product). FETCH_SIZE/WRITE_SIZE are in KB; "
                "FETCH_SIZE must be doubled on gfx950 (MI355X_MICROARCH.md, HBM section): bench.py reports traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes. "
                "TCC_EA0_RDREQ_DRAM_sum counts the read requests that went to DRAM (the rest of TCC_EA0_RDREQ_sum was served by the Infinity Cache).")
json.dump(agg, open(out + "/summary/pmc_k_conv_tiled_c2.json", "w"), indent=1)
print(json.dumps({k: v for k, v in agg.items() if k not in ("_note", "workload")}, indent=0)[:1500])
PY
# the same two traffic counters for the C4 workload (64^4: BASELINE configs[3]'s per-GPU shape), so that its bench line carries `traffic`
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/gft_pmc_c4_$grp
  timeout 900 rocprofv3 --pmc $grp --output-format csv -d /tmp/gft_pmc_c4_$grp -o pmc -- python3 "$ROOT/bench.py" --workload c4 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-clock > "$OUT/pmc_c4_$grp.log" 2>&1
  f=$(find /tmp/gft_pmc_c4_$grp -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/pmc_c4_$grp.csv"
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
agg = {}
for f in sorted(glob.glob(out + "/pmc_c4_*.csv")):
    per = {}
    for r in csv.DictReader(open(f)):
        if "k_conv_tiled" not in r["Kernel_Name"]:
            continue
        per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for c, d in per.items():
        v = list(d.values())
        agg[c] = {"per_launch_mean": sum(v) / len(v), "launches": len(v)}
agg["workload"] = "c4"
agg["_note"] = ("rocprofv3 --pmc passes (separate runs) of `python3 bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e` on MI355X; per-launch sums "
                "for the k_conv_tiled dispatches (64^4 product).  KB; FETCH_SIZE is doubled by bench.py as for c2.")
json.dump(agg, open(out + "/summary/pmc_k_conv_tiled_c4.json", "w"), indent=1)
print(json.dumps({k: v for k, v in agg.items() if k not in ("_note", "workload")}))
PY
cd "$ROOT"
cp "$OUT/summary/pmc_k_conv_tiled_c4.json" "$OUT/summary/pmc_k_conv_tiled_c2.json" profiles/r06/ 2>/dev/null  # (bench.py reads the committed files: make this run's visible to the two bench lines below)
python3 bench.py --steps 20 --warmup 3 > "$OUT/summary/bench_c2_n1.json" 2> "$OUT/bench_c2.err"
python3 bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-clock > "$OUT/summary/bench_c4_n1.json" 2> "$OUT/bench_c4.err"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench_fp64.hip -o /tmp/microbench_fp64 2> "$OUT/mb.err" && /tmp/microbench_fp64 > "$OUT/summary/microbench_fp64.txt" 2>&1
python3 tools/bench_interval.py > "$OUT/summary/interval_product.txt" 2> "$OUT/iv.err"
python3 tools/bench_interval_shapes.py > "$OUT/summary/interval_shapes.txt" 2> "$OUT/ivs.err"
python3 tools/xover_host.py > "$OUT/summary/xover_host.txt" 2> "$OUT/xover.err"
python3 tools/bench_e2e.py --limit 100 --runs 2 --bounds --only approx --gpu-only > "$OUT/e2e_bounds.log" 2>&1; tail -1 "$OUT/e2e_bounds.log" > "$OUT/summary/e2e_neurips_limit100_bounds.json"
python3 tools/bench_streaming.py 384 > "$OUT/summary/streaming_384.json" 2> "$OUT/streaming.err"
python3 tools/bench_staged.py > "$OUT/staged.log" 2>&1; cp gpurun_out/staged_vs_naive.json "$OUT/summary/" 2>/dev/null
python3 tools/sweep_tiled.py > "$OUT/summary/tiled_size_sweep.txt" 2> "$OUT/sweep.err"
python3 tools/bench_sync.py > "$OUT/summary/host_round_trip.txt" 2> "$OUT/sync.err"
python3 tools/bench_horner.py > "$OUT/summary/horner_loop.txt" 2> "$OUT/horner.err"
python3 tools/bench_shallow.py > "$OUT/summary/shallow_products.txt" 2> "$OUT/shallow.err"
python3 tools/bench_recurrence.py 32x32x32 64x64x64 65x65x65 72x72x72 96x96x96 128x128x128 24x24x24x24 6x5x100x70 200x200 400x400 1000x32 100x64 > "$OUT/summary/recurrences.txt" 2> "$OUT/rec2.err"
python3 tools/rank_emulation.py 64 4 3 > "$OUT/summary/c4_rank_emulation.txt" 2> "$OUT/rank.err"
for prog in three_populations:100 four_populations:24; do
  n=${prog%%:*}; l=${prog##*:}
  (cd /tmp && rm -rf /tmp/gft_e2e_$n && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gft_e2e_$n -o kt -- python3 "$ROOT/tools/run_sgcl.py" bench/$n.sgcl "--limit $l" 1 > "$OUT/e2e_${n}_trace.log" 2>&1; cp "$(find /tmp/gft_e2e_$n -name '*kernel_stats.csv' | head -1)" "$OUT/summary/e2e_${n}_kernel_stats.csv")
done
python3 tools/bench_div2d.py 32 64 > "$OUT/summary/div2d_slab.txt" 2> "$OUT/div2d.err"
(cd /tmp && rm -rf /tmp/gft_e2e_mix && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gft_e2e_mix -o kt -- python3 "$ROOT/tools/bench_e2e.py" --limit 100 --runs 1 --only mixture --gpu-only > "$OUT/e2e_mix_trace.log" 2>&1; cp "$(find /tmp/gft_e2e_mix -name '*kernel_stats.csv' | head -1)" "$OUT/summary/e2e_mixture_kernel_stats.csv")
(cd /tmp && rm -rf /tmp/gft_e2e_mixb && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gft_e2e_mixb -o kt -- python3 "$ROOT/tools/bench_e2e.py" --limit 100 --runs 1 --only mixture --gpu-only --bounds > "$OUT/e2e_mixb_trace.log" 2>&1; cp "$(find /tmp/gft_e2e_mixb -name '*kernel_stats.csv' | head -1)" "$OUT/summary/e2e_mixture_bounds_kernel_stats.csv")
for prog in hmm; do
  (cd /tmp && rm -rf /tmp/gft_e2e_$prog && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gft_e2e_$prog -o kt -- python3 "$ROOT/tools/bench_e2e.py" --limit 100 --runs 1 --only $prog --gpu-only > "$OUT/e2e_${prog}_trace.log" 2>&1; cp "$(find /tmp/gft_e2e_$prog -name '*kernel_stats.csv' | head -1)" "$OUT/summary/e2e_${prog}_kernel_stats.csv")
  (cd /tmp && rm -rf /tmp/gft_e2e_${prog}b && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gft_e2e_${prog}b -o kt -- python3 "$ROOT/tools/bench_e2e.py" --limit 100 --runs 1 --only $prog --gpu-only --bounds > "$OUT/e2e_${prog}b_trace.log" 2>&1; cp "$(find /tmp/gft_e2e_${prog}b -name '*kernel_stats.csv' | head -1)" "$OUT/summary/e2e_${prog}_bounds_kernel_stats.csv")
done
GFT_BATCH=0 python3 tools/bench_e2e.py --limit 100 --runs 3 --only approx --gpu-only > "$OUT/e2e_batch_off.log" 2>&1; grep -v '^{' "$OUT/e2e_batch_off.log" > "$OUT/summary/e2e_neurips_limit100_batch_off.txt"
GFT_BATCH=0 python3 tools/bench_e2e.py --limit 100 --runs 2 --only approx --gpu-only --bounds > "$OUT/e2e_bounds_batch_off.log" 2>&1; grep -v '^{' "$OUT/e2e_bounds_batch_off.log" > "$OUT/summary/e2e_neurips_limit100_bounds_batch_off.txt"
GFT_TRACE_SCANS=1 python3 tools/bench_e2e.py --limit 100 --runs 1 --only approx/hmm --gpu-only --bounds 2>&1 | grep "gft scans" | sort > "$OUT/summary/e2e_hmm_bounds_scans.txt"
python3 tools/profile_host.py approx/mixture "--limit 100" 8 > "$OUT/host_profile.log" 2>&1; python3 tools/symbolize_samples.py gpurun_out/host_profile_mixture.samples > "$OUT/summary/host_profile_mixture.txt" 2>&1
python3 tools/profile_host.py switchpoint "--limit 100 --bounds" 3 > "$OUT/host_profile_swb.log" 2>&1; python3 tools/symbolize_samples.py gpurun_out/host_profile_switchpoint_bounds.samples 30 > "$OUT/summary/host_profile_switchpoint_bounds.txt" 2>&1
python3 tools/profile_host.py switchpoint "--limit 100" 5 > "$OUT/host_profile_sw.log" 2>&1; python3 tools/symbolize_samples.py gpurun_out/host_profile_switchpoint.samples 30 > "$OUT/summary/host_profile_switchpoint.txt" 2>&1
python3 tools/profile_host.py approx/hmm "--limit 100" 30 > "$OUT/host_profile_hmm.log" 2>&1; python3 tools/symbolize_samples.py gpurun_out/host_profile_hmm.samples 30 > "$OUT/summary/host_profile_hmm.txt" 2>&1
python3 tools/bench_e2e.py --limit 100 --runs 3 > "$OUT/e2e.log" 2>&1; tail -1 "$OUT/e2e.log" > "$OUT/summary/e2e_neurips_limit100.json"
grep -v '^{' "$OUT/e2e.log" > "$OUT/summary/e2e_neurips_limit100.txt"
grep -v '^{' "$OUT/e2e_bounds.log" > "$OUT/summary/e2e_neurips_limit100_bounds.txt"
ls -la "$OUT/summary"
