#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_horner_shapes_gpu.py tests/test_interval_pins.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_e2e_snapshots.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python tools/bench_e2e.py --gpu-only --runs 5 --bounds --only approx/ 2>&1 | grep -v '^{' | cut -c1-200
} > gpurun_out/r6ab.log 2>&1
grep -v amdgpu gpurun_out/r6ab.log
