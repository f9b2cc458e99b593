#!/bin/bash
# Same-box A/B of two libraries on the small-grid points (one 5 s clip, 4 x 20 s, 8 x 20 s) and the headline:
#   bash tools/ab_small.sh TAG libA.so libB.so
TAG=$1; shift
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/$TAG; mkdir -p $OUT
one() {  # name lib args...
  local name=$1 lib=$2; shift 2
  ( GIGAAM_HIP_LIB=$lib timeout 300 python bench.py "$@" --cpu-utts 0 ) 2> $OUT/$name.err | grep -a '^{' > $OUT/$name.json
  python - "$OUT/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print(sys.argv[2], d["ms_per_step"], "ms", d["value"], d.get("kernel_classes_ms_per_step"), d.get("gpu_ids_identical"))
PY
}
i=0
for lib in "$@"; do
  for rep in 1 2; do
    one c1_${i}_$rep $lib --config 1 --steps 50 --warmup 10 --no-profile
    one b4_${i}_$rep $lib --batch 4 --steps 30 --warmup 8 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power --no-profile
  done
  one b8_$i $lib --batch 8 --steps 30 --warmup 8 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power --no-profile
  one b4prof_$i $lib --batch 4 --steps 20 --warmup 5 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power
  one head_$i $lib --steps 20 --warmup 5 --no-f32-leg --no-h2d-leg --no-f16-leg
  i=$((i+1))
done
