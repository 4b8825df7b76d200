#!/bin/bash
# A/B of two builds of the host interpreter on one box: bash tools/ab_host_lib.sh <other libgfhost.so> [repeats]
OTHER=$1; REP=${2:-2}; P=neurips2023/approx
one() { python3 tools/run_sgcl.py $P/$1/$1.sgcl "--limit 100" 10 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["program"].split("/")[-1], d["gpu_s"], d["op_stats_per_run"]["launches"], d["op_stats_per_run"]["host_tier_ops"])'; }
for i in $(seq $REP); do
  for prog in mixture hmm; do
    echo "this build   $(one $prog)"
    echo "other build  $(GENFER_HOST_LIB=$OTHER one $prog)"
  done
done
