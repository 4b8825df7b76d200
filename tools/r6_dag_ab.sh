#!/bin/bash
# first run of the launch graph: correctness subset + e2e A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_horner_shapes_gpu.py -x -q 2>&1 | tail -5
timeout 900 python -m pytest tests/test_e2e_snapshots.py -x -q -m gpu -k "mixture or hmm or limit100" 2>&1 | tail -5
for b in 1 0; do
  echo "== GFT_BATCH=$b f64"
  GFT_BATCH=$b timeout 600 python tools/bench_e2e.py --gpu-only --runs 5 --only approx/ 2>&1 | grep -v '^{' 
  echo "== GFT_BATCH=$b bounds"
  GFT_BATCH=$b timeout 900 python tools/bench_e2e.py --gpu-only --runs 3 --bounds --only approx/ 2>&1 | grep -v '^{'
done
} > gpurun_out/r6a.log 2>&1
tail -60 gpurun_out/r6a.log
