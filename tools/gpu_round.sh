#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -k "rnnt" ) > gpurun_out/pytest_gpu.log 2>&1
( timeout 300 python bench.py --steps 3 --warmup 1 --cpu-utts 0 --model v2_rnnt ) > gpurun_out/bench_rnnt.log 2>&1
tail -3 gpurun_out/pytest_gpu.log
for f in bench_rnnt; do tail -1 gpurun_out/$f.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline())
    print('$f', d['value'], d['ms_per_step'], d.get('kernel_classes_ms_per_step'), d.get('tokens_decoded_per_step'))
except Exception as e: print('$f', 'ERR', e)
"; done
