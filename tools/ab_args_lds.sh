#!/bin/bash
# Same-box A/B of GFT_ARGS_LDS (kernel arguments copied to LDS by the workgroup) on the end-to-end programs,
# alternating runs, plus the kernel trace of hmm / mixture with the default.  Run from the repo root on an MI355X box.
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/argslds; mkdir -p "$OUT"; rm -f "$OUT"/*.jsonl
for v in 1 0 1 0; do
  GFT_ARGS_LDS=$v timeout 300 python3 tools/bench_e2e.py --limit 100 --runs 5 --only approx --gpu-only 2> "$OUT/err_$v.log" | tail -1 >> "$OUT/f64_args_lds_$v.jsonl"
done
for v in 1 0; do
  GFT_ARGS_LDS=$v timeout 300 python3 tools/bench_e2e.py --limit 100 --runs 3 --only approx --bounds --gpu-only 2>> "$OUT/err_$v.log" | tail -1 >> "$OUT/bounds_args_lds_$v.jsonl"
done
for prog in hmm mixture; do
  (cd /tmp && rm -rf /tmp/gft_al_$prog && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gft_al_$prog -o kt -- python3 "$ROOT/tools/bench_e2e.py" --limit 100 --runs 1 --only approx/$prog --gpu-only > "$OUT/${prog}_trace.log" 2>&1; cp "$(find /tmp/gft_al_$prog -name '*kernel_stats.csv' | head -1)" "$OUT/e2e_${prog}_kernel_stats.csv")
done
ls -la "$OUT"
