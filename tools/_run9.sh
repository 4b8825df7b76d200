mkdir -p gpurun_out/r03
export HIP_FORCE_DEV_KERNARG=1
timeout 900 python -m pytest tests/test_dist_abi.py tests/test_dist_gpu.py tests/test_abi_symbols.py -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | head
python bench.py --steps 5 --warmup 2 > gpurun_out/r03/bench_now.json 2> gpurun_out/r03/bench_now.err; echo rc=$?
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03/bench_now.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["bound"])
for k,v in d["e2e"]["programs"].items(): print(k, v)
PY
