#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for f in build_variants/lib_*.so; do GIGAAM_HIP_LIB=$PWD/$f timeout 120 python tools/attn_bench.py; done > gpurun_out/attn.log 2>&1
cat gpurun_out/attn.log
