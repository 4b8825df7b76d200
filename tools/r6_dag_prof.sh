#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
python tools/bench_e2e.py --gpu-only --runs 3 --bounds --only approx/switchpoint 2>&1 | grep -v '^{'| cut -c1-200
python tools/bench_e2e.py --gpu-only --runs 3 --only approx/switchpoint 2>&1 | grep -v '^{'| cut -c1-200
timeout 1200 python -m pytest tests/test_e2e_snapshots.py -x -q -m gpu -k "switchpoint or limit100 or bounds" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "div_row_wavefront_bit_exact" 2>&1 | tail -3
} > gpurun_out/r6c.log 2>&1
grep -v amdgpu.ids gpurun_out/r6c.log | tail -30
