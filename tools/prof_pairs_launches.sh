#!/bin/bash
# per-launch durations of k_pair_sums for one interval product size under the default (bounded) cap
ROOT=$(pwd); export TMPDIR=/tmp; n=${1:-64}
cd /tmp; rm -rf /tmp/kt_pl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_pl -o kt -- python3 $ROOT/tools/bench_interval.py $n > /dev/null 2>&1
python3 - "$(find /tmp/kt_pl -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'k_pair_sums' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
g=[(r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'), r.get('Grid_Size_Y')) for r in rows]
n=len(d)//4
for i in range(0,len(d),n): print([round(x) for x in d[i:i+n]])
print(g[:n])
PY
