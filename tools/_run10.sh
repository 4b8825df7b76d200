mkdir -p gpurun_out/r03
export HIP_FORCE_DEV_KERNARG=1
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "wavefront or recurrences or binary_ops or exp_log or division" 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | head
timeout 900 python tools/bench_recurrence.py 32x32x32 64x64x64 24x24x24x24 100x64 48x48x48 > gpurun_out/r03/recurrences.txt 2>&1
grep -v amdgpu gpurun_out/r03/recurrences.txt | grep -E "div|log"
