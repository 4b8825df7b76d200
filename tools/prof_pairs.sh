#!/bin/bash
# k_pair_sums / k_pair_collect totals of one interval product size, bounded (default cap) vs one piece (cap 90 GB)
ROOT=$(pwd); export TMPDIR=/tmp; n=${1:-64}
for cap in default 90000; do
  cd /tmp; rm -rf /tmp/kt_pairs
  if [ $cap = default ]; then timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_pairs -o kt -- python3 $ROOT/tools/bench_interval.py $n > /dev/null 2>&1
  else GFT_RB_PAIRS_CAP_MB=$cap timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_pairs -o kt -- python3 $ROOT/tools/bench_interval.py $n > /dev/null 2>&1; fi
  echo "== cap $cap"
  python3 - "$(find /tmp/kt_pairs -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]: print(r["Name"][:48], r["Calls"], "total ms", float(r["TotalDurationNs"])/1e6, "avg us", float(r["AverageNs"])/1e3, "max us", float(r["MaxNs"])/1e3)
PY
  cd $ROOT
done
