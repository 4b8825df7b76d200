#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_horner_shapes_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_e2e_snapshots.py -x -q -m gpu 2>&1 | tail -3
for b in 1; do
GFT_BATCH=$b python tools/bench_e2e.py --gpu-only --runs 3 --bounds --only approx/hmm 2>&1 | grep -v '^{'| cut -c1-400
GFT_BATCH=$b python tools/bench_e2e.py --gpu-only --runs 3 --bounds --only approx/mixture 2>&1 | grep -v '^{'| cut -c1-400
GFT_BATCH=$b python tools/bench_e2e.py --gpu-only --runs 5 --only approx/hmm 2>&1 | grep -v '^{'| cut -c1-400
done
GFT_NZ_PROOFS=0 python tools/bench_e2e.py --gpu-only --runs 2 --bounds --only approx/hmm 2>&1 | grep -v '^{'| cut -c1-400
timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -k "c5_interval_slabs_vs_oracle or row_pair_88_whole" 2>&1 | tail -5
python tools/rank_emulation.py 64 4 3 > gpurun_out/c4_rank_emulation.txt 2>&1; tail -25 gpurun_out/c4_rank_emulation.txt | cut -c1-300
} > gpurun_out/r6c.log 2>&1
grep -v amdgpu.ids gpurun_out/r6c.log | tail -100
