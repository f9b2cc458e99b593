#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
( echo "== random"; GAM_SP_DBG=4 timeout 300 python tools/gemm_sp_test.py; echo "== zeros"; GAM_TEST_ZEROS=1 GAM_SP_DBG=4 timeout 300 python tools/gemm_sp_test.py ) 2>&1 | grep -v "tile@64\|wave5" > gpurun_out/sp_power.log
cat gpurun_out/sp_power.log
