#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
GFT_TRACE_API=1 timeout 600 python tools/run_sgcl.py neurips2023/approx/hmm/hmm.sgcl "--limit 100 --bounds" 1 > gpurun_out/r6hmm.log 2>&1
grep "materialised\|gpu_s" gpurun_out/r6hmm.log | cut -c1-1500
grep "gft api" gpurun_out/r6hmm.log | grep calls | sort -k6 -n -r | head -14
