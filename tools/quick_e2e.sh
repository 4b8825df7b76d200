#!/bin/bash
# A two-minute check on an MI355X box: the Horner-shape / interval-pin / e2e-snapshot tests, the approx programs at --limit 100 (f64 and --bounds), host profiles of switchpoint.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_horner_shapes_gpu.py tests/test_interval_pins.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_e2e_snapshots.py -x -q -m gpu 2>&1 | tail -3
echo "== bounds"
timeout 900 python tools/bench_e2e.py --gpu-only --runs 3 --bounds --only approx/ 2>&1 | grep -v '^{'
echo "== f64"
timeout 600 python tools/bench_e2e.py --gpu-only --runs 5 --only approx/ 2>&1 | grep -v '^{'
timeout 600 python tools/profile_host.py switchpoint "--limit 100 --bounds" 3
timeout 600 python tools/profile_host.py switchpoint "--limit 100" 5
} > gpurun_out/r6post.log 2>&1
tail -40 gpurun_out/r6post.log
