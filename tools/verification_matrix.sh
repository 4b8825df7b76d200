# The parity / fuzz / KAT / Horner-shape tests under every A/B switch of the library (one process per configuration): a
# result must not depend on which of two equivalent paths computed it.  Run on an MI355X box from the repo root; the record
# lands in gpurun_out/verification_matrix.txt (copy it to profiles/rNN/).
export HIP_FORCE_DEV_KERNARG=1
mkdir -p gpurun_out
OUT=gpurun_out/verification_matrix.txt
: > $OUT
# MATRIX=r5 runs the round-5 switches only (plus the baseline), with the e2e snapshot / --bounds goldens added (they are
# what exercises the recorded observation chains, the riders, the "no exact zero" proofs and the side streams)
if [ "${MATRIX:-all}" = "old_light" ]; then  # the earlier rounds' switches on the light test set (what round 5's recorded operations could interact with)
  LIGHT="not full_size and not c4_slabs and not whole_tensor and not register_blocked and not row_pair and not blocked_right and not div_row_wavefront and not recurrences_same_bits"
  for cfg in "GFT_DEFER=0" "GFT_ASYNC_LAUNCH=0" "GFT_HORNER_PIPE=0" "GFT_HORNER_LEAN=0" "GFT_FUSE_HORNER=0" "GFT_SHALLOW_MAX_TERMS=0" "GFT_HORNER_AHEAD=0" "GFT_HORNER_HOST_PHASE=0" \
             "GFT_HORNER_LOOP_MAX=0" "GFT_HOST_MAX_ELEMS=0" "GFT_DEFER=0 GFT_ASYNC_LAUNCH=0 GFT_HORNER_PIPE=0 GFT_SHALLOW_MAX_TERMS=0 GFT_HORNER_AHEAD=0 GFT_HORNER_HOST_PHASE=0"; do
    res=$(env $cfg timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_horner_shapes_gpu.py tests/test_e2e_snapshots.py -m gpu -q -x -k "$LIGHT" 2>&1 | grep -E "passed|failed" | tail -1)
    echo "$cfg : $res" | tee -a $OUT
  done
  exit 0
fi
if [ "${MATRIX:-all}" = "r5b" ]; then  # the switches added late in round 5: the row-pair grid / lanes / regime collect, the shallow kernel's pair form
  for cfg in "GFT_RB_PAIRS_LANES=0" "GFT_RB_PAIRS_LANES=1" "GFT_RB_PAIRS_CAP_MB=8" "GFT_RB_PAIRS_CAP_MB=8 GFT_RB_PAIRS_LANES=1" "GFT_RB_PAIRS_CAP_MB=90000"; do
    res=$(env $cfg timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x -k "register_blocked or row_pair or reference_order or c5 or fuzz" 2>&1 | grep -E "passed|failed" | tail -1)
    echo "$cfg : $res (interval / reference-order product tests + fuzz)" | tee -a $OUT
  done
  for cfg in "GFT_SHALLOW_PAIR_MIN=-2" "GFT_SHALLOW_PAIR_MIN=0"; do
    res=$(env $cfg timeout 1500 python -m pytest tests/test_fuzz_gpu.py tests/test_horner_shapes_gpu.py tests/test_e2e_snapshots.py -m gpu -q -x 2>&1 | grep -E "passed|failed" | tail -1)
    echo "$cfg : $res (fuzz + Horner shapes + e2e snapshots)" | tee -a $OUT
  done
  exit 0
fi
if [ "${MATRIX:-all}" = "r5c" ]; then  # the host-runtime switches of late round 5: arguments through LDS, small-block lists, the interpreter's kept chains
  LIGHT="not full_size and not c4_slabs and not whole_tensor and not register_blocked and not row_pair and not blocked_right and not div_row_wavefront and not recurrences_same_bits"
  for cfg in "GFT_ARGS_LDS=0" "GFT_SMALL_ALLOC=0" "GFH_CHAIN_TABLE=0" "GFT_ARGS_LDS=0 GFT_SMALL_ALLOC=0 GFH_CHAIN_TABLE=0"; do
    res=$(env $cfg timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_horner_shapes_gpu.py tests/test_e2e_snapshots.py -m gpu -q -x -k "$LIGHT" 2>&1 | grep -E "passed|failed" | tail -1)
    echo "$cfg : $res" | tee -a $OUT
  done
  exit 0
fi
if [ "${MATRIX:-all}" = "r5" ]; then
  LIGHT="not full_size and not c4_slabs and not whole_tensor and not register_blocked and not row_pair and not blocked_right and not div_row_wavefront and not recurrences_same_bits"
  for cfg in "GFT_BASELINE=1" "GFT_LAZY_OBSERVE=0" "GFT_OBS_RIDERS=0" "GFT_LAZY_SUM=0" "GFT_LAZY_HORNER=0" "GFT_HORNER_RIDERS=0" "GFT_NZ_PROOFS=0" "GFT_CONV_LINE=0" "GFT_SIDE_STREAMS=4" \
             "GFT_LAZY_OBSERVE=0 GFT_LAZY_SUM=0 GFT_LAZY_HORNER=0 GFT_NZ_PROOFS=0 GFT_CONV_LINE=0"; do
    res=$(env $cfg timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_horner_shapes_gpu.py tests/test_e2e_snapshots.py -m gpu -q -x -k "$LIGHT" 2>&1 | grep -E "passed|failed" | tail -1)
    echo "$cfg : $res" | tee -a $OUT
  done
  # the product / recurrence switches: the tests that exercise them
  for cfg in "GFT_RB_PAIRS_CAP_MB=8" "GFT_RB_PAIRS_CAP_MB=8 GFT_RB_PAIRS_LANES=1" "GFT_RB_PAIRS_CAP_MB=90000" "GFT_DIV_RIGHT=1"; do
    res=$(env $cfg timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x -k "register_blocked or row_pair or blocked_right or div_row_wavefront or recurrences_same_bits or fuzz" 2>&1 | grep -E "passed|failed" | tail -1)
    echo "$cfg : $res (product / recurrence tests + fuzz)" | tee -a $OUT
  done
  exit 0
fi
for cfg in "GFT_BASELINE=1" "GFT_DEFER=0" "GFT_ASYNC_LAUNCH=0" "GFT_HORNER_PIPE=0" "GFT_HORNER_LEAN=0" "GFT_DIV_WAVEFRONT=0" "GFT_DWF_DIAG=0" "GFT_DWF_PACK=0" \
           "GFT_CONV_RB_MIN_MACS=0" "GFT_CONV_RB=0" "GFT_TILED_WG_MULT=1" \
           "GFT_SHALLOW_MAX_TERMS=0" "GFT_SHALLOW_MAX_TERMS=64" "GFT_ROWS_WAVEFRONT=0" "GFT_HORNER_AHEAD=0" "GFT_HORNER_HOST_PHASE=0" "GFT_TILED_INPLACE=0" \
           "GFT_DWF_QUAD=0" "GFT_DWF_QUAD=2" "GFT_RB_PAIRS=0" "GFT_RB_PAIRS=2" "GFT_PAIRS_FIRST=0" \
           "GFT_DEFER=0 GFT_ASYNC_LAUNCH=0 GFT_HORNER_PIPE=0 GFT_DIV_WAVEFRONT=0 GFT_CONV_RB=0 GFT_SHALLOW_MAX_TERMS=0 GFT_ROWS_WAVEFRONT=0 GFT_HORNER_AHEAD=0 GFT_HORNER_HOST_PHASE=0 GFT_TILED_INPLACE=0 GFT_DWF_QUAD=0 GFT_RB_PAIRS=0 GFT_PAIRS_FIRST=0"; do
  res=$(env $cfg timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_exact_kats.py tests/test_horner_shapes_gpu.py tests/test_reference_unit_vectors.py tests/test_interval_pins.py -m gpu -q -x 2>&1 | grep -E "passed|failed" | tail -1)
  echo "$cfg : $res" | tee -a $OUT
done
