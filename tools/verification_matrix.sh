export HIP_FORCE_DEV_KERNARG=1
mkdir -p gpurun_out/r03
: > gpurun_out/r03/verification_matrix.txt
for cfg in "GFT_DEFER=0" "GFT_ASYNC_LAUNCH=0" "GFT_HORNER_PIPE=0" "GFT_HORNER_LEAN=0" "GFT_DIV_WAVEFRONT=0" "GFT_DWF_DIAG=0" "GFT_DWF_PACK=0" "GFT_CONV_RB_MIN_MACS=0" "GFT_CONV_RB=0" "GFT_TILED_WG_MULT=1" "GFT_DEFER=0 GFT_ASYNC_LAUNCH=0 GFT_HORNER_PIPE=0 GFT_DIV_WAVEFRONT=0 GFT_CONV_RB=0"; do
  res=$(env $cfg timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_exact_kats.py tests/test_horner_shapes_gpu.py tests/test_reference_unit_vectors.py tests/test_interval_pins.py -m gpu -q -x 2>&1 | grep -E "passed|failed" | tail -1)
  echo "$cfg : $res" | tee -a gpurun_out/r03/verification_matrix.txt
done
