#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -k "div_row_wavefront_bit_exact" 2>&1 | tail -5
timeout 600 python tools/bench_recurrence.py 64x64x64 65x65x65 72x72x72 96x96x96 128x128x128 10x70x130 6x5x100x70 2>&1 | grep -v amdgpu
} > gpurun_out/r6seg.log 2>&1
tail -40 gpurun_out/r6seg.log
