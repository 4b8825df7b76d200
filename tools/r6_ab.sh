#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for rep in 1 2; do
for v in 3 1 0; do
  echo "== GFT_SETTLE_REC=$v"
  GFT_SETTLE_REC=$v timeout 600 python tools/bench_e2e.py --gpu-only --runs 5 --bounds --only approx/hmm 2>&1 | grep -v '^{' | cut -c1-200
  GFT_SETTLE_REC=$v timeout 600 python tools/bench_e2e.py --gpu-only --runs 10 --only approx/hmm 2>&1 | grep -v '^{' | cut -c1-200
  GFT_SETTLE_REC=$v timeout 600 python tools/bench_e2e.py --gpu-only --runs 5 --only approx/two 2>&1 | grep -v '^{' | cut -c1-200
done
done
} > gpurun_out/r6ab.log 2>&1
grep -v amdgpu gpurun_out/r6ab.log
