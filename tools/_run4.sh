set -x
mkdir -p gpurun_out/r03
export HIP_FORCE_DEV_KERNARG=1
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
for p in mixture hmm; do
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$p -o kt -- python3 $ROOT/tools/bench_e2e.py --gpu-only --runs 2 --only approx/$p/ > $ROOT/gpurun_out/r03/kt2_$p.log 2>&1
python3 - /tmp/kt_$p $ROOT/gpurun_out/r03/gaps_$p.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last run only: take the second half by count
n = len(rows) // 2
rows = rows[n:]
st = [int(r["Start_Timestamp"]) for r in rows]; en = [int(r["End_Timestamp"]) for r in rows]
span = en[-1] - st[0]; busy = sum(e - s for s, e in zip(st, en))
gaps = [st[i + 1] - en[i] for i in range(len(rows) - 1)]
import statistics
out = open(sys.argv[2], "w")
print("kernels", len(rows), "span_ms", span / 1e6, "busy_ms", busy / 1e6, "gap_ms", sum(g for g in gaps if g > 0) / 1e6, file=out)
gs = sorted(gaps)
for q in (0.1, 0.25, 0.5, 0.75, 0.9, 0.99): print("gap q", q, gs[int(q * len(gs))], "ns", file=out)
big = [g for g in gaps if g > 20000]
print("gaps > 20us:", len(big), "sum_ms", sum(big) / 1e6, file=out)
mid = [g for g in gaps if 5000 < g <= 20000]
print("gaps 5-20us:", len(mid), "sum_ms", sum(mid) / 1e6, file=out)
PY
done
cd $ROOT
G=genfer_amd/csrc/host/genfer
N=tests/golden/sgcl/neurips2023/approx
for p in mixture hmm; do
  GFT_TRACE_API=1 $G --limit 100 $N/$p/$p.sgcl > gpurun_out/r03/trace2_$p.out 2> gpurun_out/r03/trace2_$p.err
  tail -2 gpurun_out/r03/trace2_$p.out
done
cat gpurun_out/r03/gaps_*.txt
