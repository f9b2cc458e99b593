#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -q -m gpu ) > gpurun_out/pytest_gpu.log 2>&1
( timeout 300 python bench.py --steps 5 --warmup 2 ) > gpurun_out/bench.log 2>&1
tail -6 gpurun_out/pytest_gpu.log
for f in bench; do tail -1 gpurun_out/$f.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline())
    print('$f', d['value'], d['ms_per_step'], d.get('roofline',{}).get('algorithmic_tflops'), d.get('kernel_classes_ms_per_step'), d.get('cpu_baseline'))
except Exception as e: print('$f', 'ERR', e)
"; done
