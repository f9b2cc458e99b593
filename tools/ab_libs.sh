#!/bin/bash
# Same-box interleaved A/B of two (or more) builds of the library at the small-grid points and the headline:
#   gpurun --timeout 900 -- 'bash tools/ab_libs.sh TAG gigaam_amd/libA.so gigaam_amd/libB.so'
TAG=$1; shift
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
C="--steps 30 --warmup 8 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power --cpu-utts 0 --no-profile"
one() {  # name lib args
  local name=$1 lib=$2; shift 2
  ( GIGAAM_HIP_LIB=$lib timeout 300 python bench.py "$@" $C ) 2> $OUT/$name.err | grep -a '^{' > $OUT/$name.json
  python -c "
import json
try:
    d=json.loads(open('$OUT/$name.json').read()); print('$name', d['ms_per_step'], 'ms', d['value'], 'x')
except Exception as e: print('$name', 'FAILED', e)"
}
for rep in 1 2; do
  i=0
  for lib in "$@"; do
    one c1_${i}_$rep $R/$lib --config 1
    one b4_${i}_$rep $R/$lib --batch 4
    one b8_${i}_$rep $R/$lib --batch 8
    one b32_${i}_$rep $R/$lib --batch 32
    i=$((i+1))
  done
done
