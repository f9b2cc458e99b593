#!/bin/bash
# r04 session 3: three LDS stages for the small tiles -- correctness (GEMM test + encoder goldens with the stage count forced),
# the (MT, NW, S, stages) sweep at the strong-scaling row counts, and the whole step at batch 4 / 8 / 1 clip / 32 with
# GAM_SP_STAGES=3 vs the default.     gpurun --timeout 1500 -- 'bash tools/r04_s3.sh r04_s3'
TAG=${1:-r04_s3}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
( GAM_SP_STAGES=3 timeout 900 python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "gemm_kernel or encoder_matches or ctc_bit_exact or fused_splitk or graph_replay" ) > $OUT/pytest_ns3.log 2>&1; echo "pytest(ns3) rc=$?"; tail -3 $OUT/pytest_ns3.log
( timeout 300 python -m pytest tests/test_hip_hardening.py -q -x -m gpu ) > $OUT/pytest_hardening.log 2>&1; echo "pytest(hardening) rc=$?"; tail -3 $OUT/pytest_hardening.log
C="--steps 30 --warmup 8 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power --cpu-utts 0 --no-profile"
one() {  # name env -- args
  local name=$1 e=$2; shift 3
  ( env $e timeout 300 python bench.py "$@" $C ) 2> $OUT/$name.err | grep -a '^{' > $OUT/$name.json
  python -c "
import json,sys
try:
    d=json.loads(open('$OUT/$name.json').read()); print('$name', d['ms_per_step'], 'ms', d['value'], 'x')
except Exception as e: print('$name', 'FAILED', e)"
}
for rep in 1 2; do
  one b4_ns2_$rep X=1 -- --batch 4
  one b4_ns3_$rep GAM_SP_STAGES=3 -- --batch 4
  one b8_ns2_$rep X=1 -- --batch 8
  one b8_ns3_$rep GAM_SP_STAGES=3 -- --batch 8
done
one c1_ns2 X=1 -- --config 1
one c1_ns3 GAM_SP_STAGES=3 -- --config 1
one b16_ns2 X=1 -- --batch 16
one b16_ns3 GAM_SP_STAGES=3 -- --batch 16
one b32_ns2 X=1 -- --batch 32
one b32_ns3 GAM_SP_STAGES=3 -- --batch 32
( timeout 900 python tools/smallm_sweep.py --calib --stages --rows=126,1004,2008,4016,8032 ) > $OUT/smallm_sweep_stages.txt 2> $OUT/smallm_sweep.err; echo "sweep rc=$?"
cp gpurun_out/smallm_sweep.json $OUT/smallm_sweep_stages.json 2>/dev/null
cut -c1-330 $OUT/smallm_sweep_stages.txt
