#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for a in 0 1; do echo "== asym $a"; GAM_SP_ASYM=$a GAM_SP_DBG=8 timeout 300 python tools/gemm_sp_test.py; done > gpurun_out/sp_test.log 2>&1
cat gpurun_out/sp_test.log
