export HIP_FORCE_DEV_KERNARG=1
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
for d in 0 1 4 8 9 13; do
rm -rf /tmp/kt_h
GFT_HORNER_DIAG=$d timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_h -o kt -- python3 $ROOT/tools/bench_horner.py 180 > /dev/null 2>&1
echo "diag $d"; python3 - "$(find /tmp/kt_h -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_horner" in r["Name"]:
        n=r["Name"]; n=n[n.find("gft::")+5:][:42]
        print(f'  {n:44s} {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
done
