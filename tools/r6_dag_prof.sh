#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_horner_shapes_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_e2e_snapshots.py -x -q -m gpu -k "mixture or hmm or limit100" 2>&1 | tail -3
for b in 1 0; do for s in 1 0; do
echo "== GFT_BATCH=$b GFH_SUBST_SHORTCUT=$s"
GFT_BATCH=$b GFH_SUBST_SHORTCUT=$s python tools/bench_e2e.py --gpu-only --runs 5 --only approx/mixture 2>&1 | grep -v '^{' | cut -c1-150
GFT_BATCH=$b GFH_SUBST_SHORTCUT=$s python tools/bench_e2e.py --gpu-only --runs 5 --only approx/hmm 2>&1 | grep -v '^{'| cut -c1-150
GFT_BATCH=$b GFH_SUBST_SHORTCUT=$s python tools/bench_e2e.py --gpu-only --runs 3 --bounds --only approx/hmm 2>&1 | grep -v '^{'| cut -c1-150
GFT_BATCH=$b GFH_SUBST_SHORTCUT=$s python tools/bench_e2e.py --gpu-only --runs 3 --bounds --only approx/mixture 2>&1 | grep -v '^{'| cut -c1-150
done; done
python tools/profile_host.py approx/mixture "--limit 100" 8
} > gpurun_out/r6b.log 2>&1
grep -v amdgpu.ids gpurun_out/r6b.log | tail -100
