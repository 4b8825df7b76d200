#!/bin/bash
# best-of-N e2e of the approx programs (f64) on one box, alternating two settings of an environment switch: quick_ab.sh NAME
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2 3; do for v in 1 0; do echo "== $1=$v"; env $1=$v python tools/bench_e2e.py --gpu-only --runs 8 --only approx/ 2>&1 | grep -v '^{' | grep "hmm\|mixture" | cut -c1-100; done; done
