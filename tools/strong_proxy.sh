#!/bin/bash
# strong-scaling proxies (batch 32/16/8/4 of 20 s on ONE GPU = the per-rank workloads of the 1/2/4/8-GPU strong points) + the single clip
TAG=${1:-r03_proxy}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/$TAG; mkdir -p $OUT
C="--steps 30 --warmup 8 --no-f32-leg --no-power --cpu-utts 0"
for b in 32 16 8 4 2; do
  ( timeout 300 python bench.py --batch $b $C ) > $OUT/b$b.json 2> $OUT/b$b.err
  echo "batch $b: $(grep -a '^{' $OUT/b$b.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d.get("kernel_classes_ms_per_step"))' 2>&1)"
done
( timeout 300 python bench.py --config 1 --steps 50 --warmup 10 --cpu-utts 0 --no-profile ) > $OUT/c1.json 2> $OUT/c1.err
echo "config1: $(grep -a '^{' $OUT/c1.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])' 2>&1)"
