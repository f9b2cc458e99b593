#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -4 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --cpu-utts 0 > gpurun_out/bench.log 2>gpurun_out/bench.err; tail -1 gpurun_out/bench.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_classes_ms_per_step'])"
