#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
GFT_TRACE_SCANS=1 python tools/bench_e2e.py --gpu-only --runs 1 --bounds --only approx/hmm 2>&1 | grep -v "instance\|zero pattern\|subst_var.acc" | cut -c1-360
} > gpurun_out/r6c.log 2>&1
grep -v amdgpu.ids gpurun_out/r6c.log | tail -70
