#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python tools/profile_host.py mixture "--limit 100 --bounds" 5 2>&1 | tail -1
timeout 600 python tools/profile_host.py hmm "--limit 100 --bounds" 5 2>&1 | tail -1
