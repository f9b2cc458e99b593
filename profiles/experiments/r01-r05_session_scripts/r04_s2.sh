#!/bin/bash
# r04 session 2: GPU tests of the ADVICE fixes + where the copyBuffer dispatches come from + the per-workgroup GEMM timeline
# at the strong-scaling point (M = 2008).   gpurun --timeout 1500 -- 'bash tools/r04_s2.sh r04_s2'
TAG=${1:-r04_s2}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
export GAM_TEST_REPORT=$OUT/measured_errors.jsonl
( time timeout 1200 python -m pytest tests -q -x -m gpu ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
C="--steps 10 --warmup 3 --cpu-utts 0 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power --no-profile"
( GAM_GRAPH_DEBUG=1 timeout 300 python bench.py --batch 4 $C ) > $OUT/b4_graphdebug.log 2>&1; grep -a "graph replays" $OUT/b4_graphdebug.log
cd /tmp
for v in default nograph; do
  E="X=1"; [[ $v == nograph ]] && E="GAM_GRAPH=0"
  ( env $E timeout 300 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $OUT/pf_$v -o b -- python $R/bench.py --batch 4 $C ) > $OUT/pf_$v.log 2>&1
  python $R/tools/copybuffer_origin.py $OUT/pf_$v > $OUT/copybuffer_origin_$v.txt 2>&1
  echo "== $v"; cat $OUT/copybuffer_origin_$v.txt | head -45
  rm -rf $OUT/pf_$v
done
cd $R
if [ -f gigaam_amd/libgigaam_hip_instr.so ]; then
  ( GIGAAM_HIP_LIB=$R/gigaam_amd/libgigaam_hip_instr.so GAM_SP_TLOG=1 GAM_SP_DBG=16 timeout 300 python tools/gemm_sp_test.py "2008,768,768,0;2008,3072,768,1;2008,768,3072,0;2008,2304,768,0;2008,1536,768,0;4016,768,768,0;4016,3072,768,1" ) > $OUT/gemm_timeline_m2008.txt 2>&1
  grep -a -A9 "tlog\]" $OUT/gemm_timeline_m2008.txt | tail -150
fi
