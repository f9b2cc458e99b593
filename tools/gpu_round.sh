#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for i in 1 2 3; do timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -1; done > gpurun_out/flaky.log 2>&1
cat gpurun_out/flaky.log
