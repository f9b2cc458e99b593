#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -q -m gpu ) > gpurun_out/pytest_gpu.log 2>&1
( timeout 300 python bench.py --steps 5 --warmup 2 ) > gpurun_out/bench.log 2>&1
( GAM_F16_BK=64 timeout 300 python bench.py --steps 5 --warmup 2 --cpu-utts 0 ) > gpurun_out/bench_bk64.log 2>&1
( timeout 300 python bench.py --steps 5 --warmup 2 --cpu-utts 0 --gemm f32 ) > gpurun_out/bench_f32.log 2>&1
tail -25 gpurun_out/pytest_gpu.log
for f in bench bench_bk64 bench_f32; do tail -1 gpurun_out/$f.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline())
    print('$f', d['value'], d['ms_per_step'], d.get('roofline'), d.get('kernel_classes_ms_per_step'), d.get('cpu_baseline'))
except Exception as e: print('$f', 'ERR', e)
"; done
