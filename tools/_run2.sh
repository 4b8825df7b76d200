set -x
mkdir -p gpurun_out/r03
export HIP_FORCE_DEV_KERNARG=1
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_full.log 2>&1
tail -8 gpurun_out/r03/pytest_full.log
python tools/bench_horner.py > gpurun_out/r03/horner_pipe.txt 2>&1
GFT_HORNER_PIPE=0 python tools/bench_horner.py > gpurun_out/r03/horner_nopipe.txt 2>&1
cat gpurun_out/r03/horner_pipe.txt gpurun_out/r03/horner_nopipe.txt
python tools/bench_e2e.py --gpu-only --runs 5 > gpurun_out/r03/e2e_defer.txt 2>&1
GFT_DEFER=0 GFT_HORNER_PIPE=0 python tools/bench_e2e.py --gpu-only --runs 5 --only approx > gpurun_out/r03/e2e_nodefer.txt 2>&1
python tools/bench_e2e.py --gpu-only --runs 3 --only approx --bounds > gpurun_out/r03/e2e_bounds.txt 2>&1
grep -v "^{" gpurun_out/r03/e2e_defer.txt; grep -v "^{" gpurun_out/r03/e2e_nodefer.txt; grep -v "^{" gpurun_out/r03/e2e_bounds.txt
