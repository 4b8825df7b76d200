mkdir -p gpurun_out/r03
export HIP_FORCE_DEV_KERNARG=1
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_lean.log 2>&1
grep -E "passed|failed|FAILED|Error" gpurun_out/r03/pytest_lean.log | head
for i in 1 2; do timeout 600 python -m pytest tests/test_e2e_snapshots.py -k "bounds" -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED" | head -3; done
python tools/bench_e2e.py --gpu-only --runs 3 --only approx --bounds > gpurun_out/r03/e2e_bounds_lean.txt 2>&1
grep -v "^{" gpurun_out/r03/e2e_bounds_lean.txt
cd /tmp
rm -rf /tmp/kt_h
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_h -o kt -- python3 $ROOT/tools/bench_horner.py 180 > /dev/null 2>&1
python3 - "$(find /tmp/kt_h -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_horner" in r["Name"]:
        n=r["Name"]; n=n[n.find("gft::")+5:][:42]
        print(f'  {n:44s} {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
