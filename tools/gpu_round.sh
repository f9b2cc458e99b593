#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "rnnt" > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
for cfg in "--model v3_e2e_rnnt --batch 32 --seconds 12.5 --rnnt-blank-bias 18" "--model v2_rnnt --batch 32 --seconds 20"; do
  echo "== $cfg"
  timeout 300 python bench.py $cfg --cpu-utts 0 --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['tokens_decoded_per_step'], d.get('kernel_classes_ms_per_step')['decode'])"
done > gpurun_out/cfgs.log 2>&1
cat gpurun_out/cfgs.log
