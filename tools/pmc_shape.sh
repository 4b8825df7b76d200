#!/bin/bash
# Kernel trace + PMC passes of the forced-tiled product at one shape:  bash tools/pmc_shape.sh 64x64x64 [out_dir]
# (separate rocprofv3 runs: the pool forbids mixing --pmc with trace domains).  Prints per-launch means of k_conv_tiled.
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
SH=$1
OUT=${2:-$ROOT/gpurun_out/pmc_$SH}
mkdir -p "$OUT"
cd /tmp
rm -rf /tmp/pk_$SH; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk_$SH -o kt -- python3 "$ROOT/tools/sweep_tiled.py" $SH > "$OUT/kt.log" 2>&1
cp "$(find /tmp/pk_$SH -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
n=0
for grp in "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  n=$((n+1))
  rm -rf /tmp/pp_${SH}_$n; timeout 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/pp_${SH}_$n -o pmc -- python3 "$ROOT/tools/sweep_tiled.py" $SH > "$OUT/pmc_$n.log" 2>&1
  f=$(find /tmp/pp_${SH}_$n -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/pmc_$n.csv"
done
python3 - "$OUT" "$SH" <<'PY'
import csv, glob, json, sys
out, sh = sys.argv[1], sys.argv[2]
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
        agg[c] = sum(v) / len(v)
for r in csv.DictReader(open(out + "/kernel_stats.csv")):
    agg["ns:" + r["Name"].split("(")[0][-40:]] = float(r["AverageNs"])
agg["shape"] = sh
json.dump(agg, open(out + "/summary.json", "w"), indent=1)
for k, v in agg.items():
    print(f"{k:48s} {v if isinstance(v, str) else format(v, '.4g')}")
PY
