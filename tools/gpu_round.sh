#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "vad or emotion or api" > gpurun_out/pytest_gpu.log 2>&1; tail -12 gpurun_out/pytest_gpu.log
timeout 900 python tools/bench_configs.py > gpurun_out/configs.log 2>&1; tail -12 gpurun_out/configs.log
