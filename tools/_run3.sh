set -x
mkdir -p gpurun_out/r03
export HIP_FORCE_DEV_KERNARG=1
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
for v in 1 0; do
  GFT_HORNER_PIPE=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_h$v -o kt -- python3 $ROOT/tools/bench_horner.py 180 > $ROOT/gpurun_out/r03/kt_horner_$v.log 2>&1
  cp "$(find /tmp/kt_h$v -name '*kernel_stats.csv' | head -1)" $ROOT/gpurun_out/r03/horner180_kernel_stats_pipe$v.csv
done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_mb -o kt -- python3 $ROOT/tools/bench_e2e.py --gpu-only --runs 1 --only mixture --bounds > $ROOT/gpurun_out/r03/kt_mixture_bounds.log 2>&1
cp "$(find /tmp/kt_mb -name '*kernel_stats.csv' | head -1)" $ROOT/gpurun_out/r03/e2e_mixture_bounds_kernel_stats.csv
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_m -o kt -- python3 $ROOT/tools/bench_e2e.py --gpu-only --runs 1 --only mixture > $ROOT/gpurun_out/r03/kt_mixture.log 2>&1
cp "$(find /tmp/kt_m -name '*kernel_stats.csv' | head -1)" $ROOT/gpurun_out/r03/e2e_mixture_kernel_stats.csv
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_hm -o kt -- python3 $ROOT/tools/bench_e2e.py --gpu-only --runs 1 --only hmm > $ROOT/gpurun_out/r03/kt_hmm.log 2>&1
cp "$(find /tmp/kt_hm -name '*kernel_stats.csv' | head -1)" $ROOT/gpurun_out/r03/e2e_hmm_kernel_stats.csv
cd $ROOT
python -m pytest tests -m gpu -q --durations=25 -k "e2e_snapshots" > gpurun_out/r03/pytest_dur.log 2>&1
tail -40 gpurun_out/r03/pytest_dur.log
