#!/bin/bash
# Per-launch timeline (start / end of every kernel) of one end-to-end program: bash tools/trace_timeline.sh mixture ["--bounds"]
export TMPDIR=/tmp
ROOT=$(pwd); prog=$1; extra=$2; OUT=$ROOT/gpurun_out/timeline; mkdir -p "$OUT"
(cd /tmp && rm -rf /tmp/gft_tl && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gft_tl -o kt -- python3 "$ROOT/tools/bench_e2e.py" --limit 100 --runs 3 --only approx/$prog --gpu-only $extra > "$OUT/${prog}_trace.log" 2>&1
 f=$(find /tmp/gft_tl -name '*kernel_trace.csv' | head -1)
 python3 - "$f" "$OUT/${prog}_timeline.csv.gz" <<'P'
import csv, gzip, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with gzip.open(sys.argv[2], "wt") as g:
    g.write("start_ns,end_ns,kernel,grid,wg\n")
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gft::", "")
        g.write(f'{int(r["Start_Timestamp"]) - t0},{int(r["End_Timestamp"]) - t0},{name},{r.get("Grid_Size_X", r.get("Grid_Size", ""))},{r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))}\n')
P
)
ls -la "$OUT"
