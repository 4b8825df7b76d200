#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
GFT_TRACE_API=1 timeout 600 python tools/run_sgcl.py neurips2023/approx/switchpoint/switchpoint.sgcl "--limit 100 --bounds" 1 > gpurun_out/r6sw.log 2>&1
grep "host-tier Horner\|gpu_s" gpurun_out/r6sw.log | cut -c1-600
