export HIP_FORCE_DEV_KERNARG=1
for cfg in "GFT_HORNER_LEAN=0" "GFT_DEFER=0" "GFT_HORNER_PIPE=0" "GFT_ASYNC_LAUNCH=0"; do
  echo "== $cfg"
  env $cfg timeout 600 python -m pytest tests/test_e2e_snapshots.py -k "two_populations-bounds" -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED|vs" | cut -c1-250 | head -5
done
