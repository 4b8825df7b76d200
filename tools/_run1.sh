set -x
mkdir -p gpurun_out/r03
export HIP_FORCE_DEV_KERNARG=1
G=genfer_amd/csrc/host/genfer
N=tests/golden/sgcl/neurips2023/approx
for p in mixture hmm; do
  GFT_TRACE_API=1 $G --limit 100 $N/$p/$p.sgcl > gpurun_out/r03/trace_$p.out 2> gpurun_out/r03/trace_$p.err
  GFT_TRACE_API=1 $G --limit 100 --bounds $N/$p/$p.sgcl > gpurun_out/r03/trace_${p}_bounds.out 2> gpurun_out/r03/trace_${p}_bounds.err
done
timeout 1500 python -m pytest tests -m gpu -x -q -k "e2e_snapshots or display or full_size or 64cubed" > gpurun_out/r03/pytest_new.log 2>&1
tail -5 gpurun_out/r03/pytest_new.log
