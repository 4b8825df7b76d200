export HIP_FORCE_DEV_KERNARG=1
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_fin.log 2>&1
grep -E "passed|failed|FAILED|Error" gpurun_out/r03/pytest_fin.log | head
timeout 900 python tools/bench_interval.py 32 64 128 > gpurun_out/r03/interval_product.txt 2>&1
grep -v amdgpu gpurun_out/r03/interval_product.txt
