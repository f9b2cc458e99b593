#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -5 gpurun_out/pytest_gpu.log
for cfg in "--batch 1 --seconds 5" "--batch 4 --seconds 20" "--batch 8 --seconds 20"; do
  for sk in 1 0; do
  echo "== $cfg splitk=$sk"
  GAM_SPLITK=$sk timeout 300 python bench.py $cfg --cpu-utts 0 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('kernel_classes_ms_per_step'))"
done; done > gpurun_out/cfgs.log 2>&1
cat gpurun_out/cfgs.log
