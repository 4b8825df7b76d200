# The parity / fuzz / KAT / Horner-shape / end-to-end tests under every A/B switch the library still has (round 6: 15 environment
# switches, README), one process per configuration: a result must not depend on which of two equivalent paths computed it.
# Run on an MI355X box from the repo root; the record lands in gpurun_out/verification_matrix.txt (copy it to profiles/rNN/)
# with the wall time of every configuration.  JOBS configurations run side by side on the one GPU (default 4: the tests are
# small launches and host work, a serial matrix is 16 x 250 s); CONFIGS="A=0;B=0 C=0" runs just those (';' separated).
export HIP_FORCE_DEV_KERNARG=1
mkdir -p gpurun_out
OUT=gpurun_out/verification_matrix.txt
: > $OUT
T0=$(date +%s)
JOBS=${JOBS:-4}
# the heavy full-size tests are the default configuration's business (the full GPU suite); the matrix runs everything else
LIGHT="not full_size and not c4_slabs and not whole_tensor and not c5_interval_slabs"
ALL="GFT_BASELINE=1;GFT_BATCH=0;GFT_LAZY_OBSERVE=0;GFT_LAZY_SUM=0;GFT_LAZY_HORNER=0;GFT_NZ_PROOFS=0;GFT_DEFER=0;GFT_ASYNC_LAUNCH=0;GFT_DIV_WAVEFRONT=0;GFT_EXP_RIGHT=0;GFT_SHALLOW_MAX_TERMS=0;GFT_HORNER_LOOP_MAX=0;GFT_HOST_MAX_ELEMS=0;GFH_SUBST_SHORTCUT=0;GFH_CHAIN_TABLE=0;GFT_BATCH=0 GFT_LAZY_OBSERVE=0 GFT_LAZY_SUM=0 GFT_LAZY_HORNER=0 GFT_NZ_PROOFS=0 GFT_DEFER=0 GFT_ASYNC_LAUNCH=0 GFH_SUBST_SHORTCUT=0"
IFS=';' read -r -a CFGS <<< "${CONFIGS:-$ALL}"
one() {
  cfg="$1"
  t0=$(date +%s)
  res=$(env $cfg timeout 2400 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_exact_kats.py tests/test_horner_shapes_gpu.py tests/test_reference_unit_vectors.py \
        tests/test_interval_pins.py tests/test_e2e_snapshots.py -m gpu -q -x -p no:cacheprovider -k "$LIGHT" 2>&1 | grep -E "passed|failed|error" | tail -1)
  echo "$cfg : $res [$(( $(date +%s) - t0 )) s]"
}
running=0
for cfg in "${CFGS[@]}"; do
  one "$cfg" >> $OUT &
  running=$((running + 1))
  if [ $running -ge $JOBS ]; then
    wait -n
    running=$((running - 1))
  fi
done
wait
cat $OUT
echo "whole matrix (${#CFGS[@]} configurations, $JOBS at a time): $(( $(date +%s) - T0 )) s" | tee -a $OUT
