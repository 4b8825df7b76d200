#!/bin/bash
# PMC passes (separate rocprofv3 runs) of one python tool, summed per launch over the kernels whose name contains <substr>:
#   bash tools/pmc_kernel.sh <substr> <out_dir> tools/bench_interval.py 128
set -u
export TMPDIR=/tmp
ROOT=$(pwd)
SUB=$1; OUT=$2; shift 2
case "$OUT" in /*) ;; *) OUT=$ROOT/$OUT ;; esac
mkdir -p "$OUT"
SCRIPT=$ROOT/$1; shift
cd /tmp
n=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD"; do
  n=$((n+1))
  rm -rf /tmp/pk_$n; timeout 600 rocprofv3 --pmc $grp --output-format csv -d /tmp/pk_$n -o pmc -- python3 "$SCRIPT" "$@" > "$OUT/pmc_$n.log" 2>&1
  f=$(find /tmp/pk_$n -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/pmc_$n.csv"
done
python3 - "$OUT" "$SUB" <<'PY'
import csv, glob, json, sys
out, sub = sys.argv[1], sys.argv[2]
agg = {}
for f in sorted(glob.glob(out + "/pmc_*.csv")):
    per = {}
    for r in csv.DictReader(open(f)):
        if sub not in r["Kernel_Name"]:
            continue
        per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for c, d in per.items():
        v = list(d.values())
        agg[c] = {"per_launch_max": max(v), "per_launch_mean": sum(v) / len(v), "launches": len(v)}
json.dump(agg, open(out + "/summary.json", "w"), indent=1)
for k, v in agg.items():
    print(f"{k:28s} max {v['per_launch_max']:.4g}  mean {v['per_launch_mean']:.4g}  launches {v['launches']}")
PY
