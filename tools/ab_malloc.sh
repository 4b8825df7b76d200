#!/bin/bash
# A/B of glibc malloc tunables on the host-bound end-to-end programs (alternating processes on one box)
P=neurips2023/approx
one() { python3 tools/run_sgcl.py $P/$1/$1.sgcl '--limit 100' 10 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["program"].split("/")[-1], d["gpu_s"])'; }
for i in 1 2 3; do
  for prog in mixture hmm; do
    echo "default        $(one $prog)"
    echo "tcache_count   $(GLIBC_TUNABLES=glibc.malloc.tcache_count=60000 one $prog)"
  done
done
