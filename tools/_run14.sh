export HIP_FORCE_DEV_KERNARG=1
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_pad.log 2>&1
grep -E "passed|failed|FAILED|Error" gpurun_out/r03/pytest_pad.log | head
python tools/bench_e2e.py --gpu-only --runs 5 --only approx > gpurun_out/r03/e2e_pad.txt 2>&1
grep -v "^{" gpurun_out/r03/e2e_pad.txt | grep -v amdgpu
