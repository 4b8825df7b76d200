#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for f in 4096 0 1024 16384; do
echo "== batch_flush $f"
GFT_BATCH_FLUSH=$f python tools/bench_e2e.py --gpu-only --runs 5 --only approx/mixture 2>&1 | grep -v '^{' | cut -c1-330
GFT_BATCH_FLUSH=$f python tools/bench_e2e.py --gpu-only --runs 5 --only approx/hmm 2>&1 | grep -v '^{'| cut -c1-330
GFT_BATCH_FLUSH=$f python tools/bench_e2e.py --gpu-only --runs 3 --bounds --only approx/hmm 2>&1 | grep -v '^{'| cut -c1-330
GFT_BATCH_FLUSH=$f python tools/bench_e2e.py --gpu-only --runs 3 --bounds --only approx/mixture 2>&1 | grep -v '^{'| cut -c1-330
done
timeout 900 python -m pytest tests/test_horner_shapes_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_e2e_snapshots.py -x -q -m gpu 2>&1 | tail -3
} > gpurun_out/r6c.log 2>&1
grep -v amdgpu.ids gpurun_out/r6c.log | tail -40
