set -x
mkdir -p gpurun_out/r03
export HIP_FORCE_DEV_KERNARG=1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_async.log 2>&1
tail -5 gpurun_out/r03/pytest_async.log
python tools/bench_e2e.py --gpu-only --runs 5 --only approx > gpurun_out/r03/e2e_async.txt 2>&1
GFT_ASYNC_LAUNCH=0 python tools/bench_e2e.py --gpu-only --runs 5 --only approx > gpurun_out/r03/e2e_sync.txt 2>&1
python tools/bench_e2e.py --gpu-only --runs 3 --only approx --bounds > gpurun_out/r03/e2e_bounds_async.txt 2>&1
grep -v "^{" gpurun_out/r03/e2e_async.txt gpurun_out/r03/e2e_sync.txt gpurun_out/r03/e2e_bounds_async.txt
python bench.py --steps 5 --warmup 2 > gpurun_out/r03/bench_async.json 2> gpurun_out/r03/bench_async.err; tail -c 1500 gpurun_out/r03/bench_async.json
