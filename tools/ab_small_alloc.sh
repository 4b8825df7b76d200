#!/bin/bash
# A/B of the small-block lists (gft_small_alloc.hpp; GFT_SMALL_ALLOC=0: operator new / delete) on the end-to-end programs,
# alternating processes on one box; best of 10 (f64) / 3 (--bounds) "Total inference time" per process.
# Usage: bash tools/ab_small_alloc.sh [repeats, default 3] [bounds: 1 = also the --bounds rows]
P=neurips2023/approx
REP=${1:-3}; BOUNDS=${2:-1}
one() { python3 tools/run_sgcl.py $P/$1/$1.sgcl "--limit 100 $2" $3 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["program"].split("/")[-1], d["flags"], d["gpu_s"])'; }
for i in $(seq $REP); do
  for prog in mixture hmm two_populations; do
    echo "lists on   $(one $prog '' 10)"
    echo "lists off  $(GFT_SMALL_ALLOC=0 one $prog '' 10)"
  done
done
if [ "$BOUNDS" = 1 ]; then
  for prog in hmm mixture; do
    echo "lists on   $(one $prog --bounds 3)"
    echo "lists off  $(GFT_SMALL_ALLOC=0 one $prog --bounds 3)"
  done
fi
