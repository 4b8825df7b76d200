export HIP_FORCE_DEV_KERNARG=1
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_final.log 2>&1
grep -E "passed|failed|FAILED|Error" gpurun_out/r03/pytest_final.log | head
timeout 900 python tools/bench_recurrence.py 32x32x32 64x64x64 200x200 24x24x24x24 400x400 48x48x48 100x64 > gpurun_out/r03/recurrences.txt 2>&1
grep -v amdgpu gpurun_out/r03/recurrences.txt
