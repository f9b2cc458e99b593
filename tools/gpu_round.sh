#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 300 python bench.py > gpurun_out/bench.log 2>gpurun_out/bench.err; tail -3 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
GAM_SP=0 timeout 300 python bench.py --no-profile > gpurun_out/bench_nosp.log 2>/dev/null; tail -1 gpurun_out/bench_nosp.log
