mkdir -p gpurun_out/r03
export HIP_FORCE_DEV_KERNARG=1
export TMPDIR=/tmp
ROOT=$(pwd)
G=$ROOT/genfer_amd/csrc/host/genfer
cd /tmp
rm -rf /tmp/kt_hb
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_hb -o kt -- python3 $ROOT/tools/bench_e2e.py --gpu-only --runs 1 --only hmm --bounds > /dev/null 2>&1
cp "$(find /tmp/kt_hb -name '*kernel_stats.csv' | head -1)" $ROOT/gpurun_out/r03/e2e_hmm_bounds_kernel_stats.csv
python3 - $ROOT/gpurun_out/r03/e2e_hmm_bounds_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
print("kernel total ms", tot/1e6, "launches", sum(int(r["Calls"]) for r in rows))
for r in rows[:14]:
    n=r["Name"]; n=n[n.find("gft::")+5:][:55]
    print(f'  {n:57s} {r["Calls"]:>6s} {int(r["TotalDurationNs"])/1e6:9.2f} ms avg {float(r["AverageNs"])/1e3:6.1f} us')
PY
cd $ROOT
GFT_TRACE_API=1 GFT_TRACE_SCANS=1 $G --limit 100 --bounds tests/golden/sgcl/neurips2023/approx/hmm/hmm.sgcl 2>&1 >/dev/null | sort -k6 -n -r | head -24
