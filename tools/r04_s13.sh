#!/bin/bash
# r04 session 13: the 6-wave 128 x 192 tile -- correctness with the tile forced (two / three stages, both arithmetic families),
# the sweep with it, whole-step A/B planned (provisional model) vs the tile excluded.
TAG=${1:-r04_s13}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for st in 2 3; do
  ( GAM_SP_MT=2 GAM_SP_NW=3 GAM_SP_STAGES=$st timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_fastmode.py -q -x -m gpu -k "gemm_kernel or encoder_matches or ctc_bit_exact or fused_splitk or graph_replay or gemm_one_term or encoder_in_speed" ) > $OUT/pytest_nw3_s$st.log 2>&1; echo "pytest(nw3, stages $st) rc=$?"; grep -a "passed\|failed" $OUT/pytest_nw3_s$st.log | tail -1
done
( timeout 900 python tools/smallm_sweep.py --calib --stages --rows=1004,2008,4016,8032 ) > $OUT/smallm_sweep_nw3.txt 2> $OUT/smallm_sweep.err; echo "sweep rc=$?"
cp gpurun_out/smallm_sweep.json $OUT/smallm_sweep_nw3.json 2>/dev/null
cut -c1-300 $OUT/smallm_sweep_nw3.txt
C="--steps 30 --warmup 8 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power --cpu-utts 0 --no-profile"
for rep in 1 2; do
  for b in 4 8 16; do
    for v in plan nw24; do
      E="X=1"; [[ $v == nw24 ]] && E="GAM_SP_NO_NW3=1"
      ( env $E timeout 300 python bench.py --batch $b $C ) 2> $OUT/b${b}_${v}_$rep.err | grep -a '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b$b $v', d['ms_per_step'])"
    done
  done
done
