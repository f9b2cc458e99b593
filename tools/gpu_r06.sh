#!/bin/bash
# r06 GPU sessions (one stage per gpurun call):  gpurun --timeout 1500 -- 'bash tools/gpu_r06.sh <stage>'
# Provenance of profiles/r06_* (index: profiles/r06_INDEX.txt).  Stages s1-s5 / s11 use VARIANT builds of the library that are not kept in the
# tree; each is the build command of gigaam_amd/build.py (hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC gigaam_amd/csrc/gam_api.hip -o ...)
# with:  libgigaam_hip.so (s1-s3: r05's flags, i.e. WITHOUT -fno-slp-vectorize)   _noslp: -fno-slp-vectorize   _audit: -DGAM_RC_AUDIT=1
#        _audit_noslp: both   _pv2: -fno-slp-vectorize -DGAM_ATT_PV_TERMS=2   _nt / _sc1: -DGAM_RC_NOL1=1 / 2 (weight loads past the L1; the switch
#        was removed again after s1 refuted it).  The canaries build from tools/*.hip with the command in their first comment block.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
STAGE=${1:-s1}
OUT=$R/gpurun_out/r06_$STAGE; mkdir -p $OUT
export TMPDIR=/tmp
repro() {  # name lib env... -- args
  local name=$1 lib=$2; shift 2
  ( env GIGAAM_HIP_LIB=$R/gigaam_amd/$lib "$@" ) 2>> $OUT/$name.err | grep -a "REPRO" | tee -a $OUT/repro.txt
}
case $STAGE in
s1)  # the co-residency mechanism: L1-hit canary + the decode reproducer with L1-bypassing weight loads
  timeout 600 python tools/l1_canary.py 6 2> $OUT/canary.err | tee $OUT/canary.txt
  repro base   libgigaam_hip.so     GAM_RNNT_EXCLUSIVE=0 timeout 300 python tools/coresidency_repro.py 2,1,4 300 gemm640
  repro nt     libgigaam_hip_nt.so  GAM_RNNT_EXCLUSIVE=0 timeout 300 python tools/coresidency_repro.py 2,1,4 300 gemm640
  repro sc1    libgigaam_hip_sc1.so GAM_RNNT_EXCLUSIVE=0 timeout 300 python tools/coresidency_repro.py 2,1,4 300 gemm640
  repro base2  libgigaam_hip.so     GAM_RNNT_EXCLUSIVE=0 timeout 300 python tools/coresidency_repro.py 2,1 300 gemm640
  repro legacy libgigaam_hip.so     GAM_RNNT_EXCLUSIVE=0 GAM_SP_MIN_M=1073741824 timeout 300 python tools/coresidency_repro.py 2,1 300 legacy
  repro attn   libgigaam_hip.so     GAM_RNNT_EXCLUSIVE=0 timeout 300 python tools/coresidency_repro.py 2 300 attention
  repro none   libgigaam_hip.so     GAM_RNNT_EXCLUSIVE=0 timeout 300 python tools/coresidency_repro.py 2 300 none
  repro excl   libgigaam_hip.so     GAM_RNNT_EXCLUSIVE=1 timeout 300 python tools/coresidency_repro.py 2,1 300 gemm640
  ;;
s2)  # C = 1 (no hand-offs at all): baseline alone, rate beside the GEMM, and the audit build (tools: gam_decode_cluster.h GAM_RC_AUDIT)
  repro c1none   libgigaam_hip.so       GAM_RNNT_EXCLUSIVE=0 timeout 300 python tools/coresidency_repro.py 1 2000 none
  repro c1gemm   libgigaam_hip.so       GAM_RNNT_EXCLUSIVE=0 timeout 300 python tools/coresidency_repro.py 1 3000 gemm640
  repro c1audit  libgigaam_hip_audit.so GAM_RNNT_EXCLUSIVE=0 timeout 600 python tools/coresidency_repro.py 1 3000 gemm640
  repro c2audit  libgigaam_hip_audit.so GAM_RNNT_EXCLUSIVE=0 timeout 600 python tools/coresidency_repro.py 2 1500 gemm640
  repro c1legacy libgigaam_hip.so       GAM_RNNT_EXCLUSIVE=0 GAM_SP_MIN_M=1073741824 timeout 300 python tools/coresidency_repro.py 1 3000 legacy
  repro c1attn   libgigaam_hip.so       GAM_RNNT_EXCLUSIVE=0 timeout 300 python tools/coresidency_repro.py 1 2000 attention
  repro c1auditn libgigaam_hip_audit.so GAM_RNNT_EXCLUSIVE=0 timeout 600 python tools/coresidency_repro.py 1 1000 none
  grep -a "gam-audit\|MISMATCH" $OUT/c1audit.err | head -150 > $OUT/c1audit_head.txt
  grep -a "gam-audit\|MISMATCH" $OUT/c2audit.err | head -80 > $OUT/c2audit_head.txt
  grep -a "gam-audit\] total\|gam-audit\] decode" $OUT/*.err | awk '{print $1, $2, $3, $4, $5, $6, $7, $8, $9, $10, $11, $12}' | sort | uniq -c | sort -rn | head -40 > $OUT/audit_summary.txt
  tail -5 $OUT/c1audit_head.txt; cat $OUT/audit_summary.txt | head -20
  ;;
s3)  # v_pk_fma_f32 low half: canary + the decode without hipcc's packed FMAs (-fno-slp-vectorize)
  timeout 900 python tools/pkfma_canary.py 4 2> $OUT/pkfma.err | tee $OUT/pkfma.txt
  repro c1noslp   libgigaam_hip_noslp.so       GAM_RNNT_EXCLUSIVE=0 timeout 300 python tools/coresidency_repro.py 1 3000 gemm640
  repro c1base    libgigaam_hip.so             GAM_RNNT_EXCLUSIVE=0 timeout 300 python tools/coresidency_repro.py 1 3000 gemm640
  repro c1audns   libgigaam_hip_audit_noslp.so GAM_RNNT_EXCLUSIVE=0 timeout 600 python tools/coresidency_repro.py 1 1500 gemm640
  repro c2audns   libgigaam_hip_audit_noslp.so GAM_RNNT_EXCLUSIVE=0 timeout 600 python tools/coresidency_repro.py 2 1000 gemm640
  repro c1noslp2  libgigaam_hip_noslp.so       GAM_RNNT_EXCLUSIVE=0 timeout 300 python tools/coresidency_repro.py 1,2,3,4 2000 gemm640
  grep -a "gam-audit\] total" $OUT/*.err | tee $OUT/audit_totals.txt
  grep -a "gam-audit\]   site" $OUT/c1audns.err | head -40 > $OUT/c1audns_head.txt
  ;;
s4)  timeout 900 python tools/pkfma_canary.py 4 2> $OUT/pkfma.err | tee $OUT/pkfma.txt; tail -3 $OUT/pkfma.err ;;
s5)  timeout 900 python tools/pkfma_rule.py 3 2> $OUT/rule.err | tee $OUT/rule.txt; tail -3 $OUT/rule.err ;;
s6)  # mid-round check: GPU tests, smoke, the driver's bench command, config 3 / 4 with and without the whole-CU claim
  export GAM_TEST_REPORT=$OUT/measured_errors.jsonl; rm -f $GAM_TEST_REPORT
  ( time timeout 1500 python -m pytest tests -q -m gpu -x ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
  ( timeout 300 python __graft_entry__.py --smoke ) > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
  ( timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 600 $OUT/bench.log
  for ex in 1 0 1 0; do
    for c in 3 4; do
      ( GAM_RNNT_EXCLUSIVE=$ex timeout 400 python bench.py --config $c --cpu-utts 0 --steps 8 --warmup 2 --no-profile --no-power ) 2>> $OUT/excl.err | grep -a '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('config $c exclusive $ex:', d['ms_per_step'], 'ms', d['value'], 'x')" | tee -a $OUT/excl.txt
    done
  done
  ;;
s7)  # packed rows: the new tests, then everything ragged
  ( timeout 900 python -m pytest tests/test_hip_varlen.py -q -x ) > $OUT/varlen.log 2>&1; echo "varlen rc=$?"; tail -25 $OUT/varlen.log
  ( timeout 1200 python -m pytest tests -q -m gpu -x -k "ragged or live or fullsize or model_api or longform" ) > $OUT/ragged.log 2>&1; echo "ragged rc=$?"; tail -8 $OUT/ragged.log
  ;;
s8)  # packed rows: perf A/B on the ragged lines (SURVEY 8d's second run of configs 2 / 3), configs 4 / 5
  ( timeout 900 python -m pytest tests/test_hip_varlen.py -q -x ) > $OUT/varlen.log 2>&1; echo "varlen rc=$?"; tail -3 $OUT/varlen.log
  C="--cpu-utts 0 --steps 12 --warmup 3 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power"
  for rep in 1 2; do
    for extra in "--config 2 --ragged --no-pack" "--config 2 --ragged" "--config 3 --ragged --no-pack" "--config 3 --ragged" "--config 4 --no-pack --steps 4" "--config 4 --steps 4" "--config 5 --no-pack --steps 4" "--config 5 --steps 4" "--config 2"; do
      ( timeout 500 python bench.py $C $extra ) 2>> $OUT/ab.err | grep -a '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$extra:', d['ms_per_step'], 'ms', d['value'], 'x', d.get('kernel_classes_ms_per_step'))" | tee -a $OUT/ab.txt
    done
  done
  ;;
s9)  # packed rows after the round-robin dealing of the stem conv's tiles
  ( timeout 900 python -m pytest tests/test_hip_varlen.py -q -x ) > $OUT/varlen.log 2>&1; echo "varlen rc=$?"; tail -3 $OUT/varlen.log
  C="--cpu-utts 0 --steps 12 --warmup 3 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power"
  for rep in 1 2; do
    for extra in "--config 2 --ragged --no-pack" "--config 2 --ragged" "--config 3 --ragged" "--config 2"; do
      ( timeout 500 python bench.py $C $extra ) 2>> $OUT/ab.err | grep -a '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$extra:', d['ms_per_step'], 'ms', d['value'], 'x', d.get('kernel_classes_ms_per_step'))" | tee -a $OUT/ab.txt
    done
  done
  ;;
s10)  # kernel traces of the ragged config-2 step, padded vs packed rows
  cd /tmp
  for v in nopack pack; do
    X=""; [ $v = nopack ] && X="--no-pack"
    ( timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/pf_$v -o b -- python $R/bench.py --config 2 --ragged $X --steps 5 --warmup 2 --cpu-utts 0 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power ) > $OUT/pf_$v.log 2>&1
    DB=$(find $OUT/pf_$v -name "*.db" | head -1)
    [ -n "$DB" ] && python $R/tools/rocpd_summary.py $DB $OUT/trace_ragged_${v}_summary.txt "rocprofv3 kernel trace of bench.py --config 2 --ragged ($v)" > /dev/null 2>&1
  done
  find $OUT -name "*.db" -delete
  head -30 $OUT/trace_ragged_pack_summary.txt
  ;;
s11)  # EXPERIMENT: two-term P.V attention (-DGAM_ATT_PV_TERMS=2) against the bars that matter: reference ids, every GPU parity test
  C="--steps 12 --warmup 3 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power"
  for lib in libgigaam_hip.so libgigaam_hip_pv2.so libgigaam_hip.so libgigaam_hip_pv2.so; do
    ( GIGAAM_HIP_LIB=$R/gigaam_amd/$lib timeout 600 python bench.py $C ) 2>> $OUT/bench.err | grep -a '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d.get('cpu_baseline', {})
print('$lib:', d['ms_per_step'], 'ms', d['value'], 'x attn', d['kernel_classes_ms_per_step'].get('attn'), 'ids vs reference', c.get('gpu_ids_identical'), 'mismatch', c.get('mismatch_logits'), c.get('mismatch_reference_min_margin'))" | tee -a $OUT/pv2.txt
  done
  ( GIGAAM_HIP_LIB=$R/gigaam_amd/libgigaam_hip_pv2.so timeout 1500 python -m pytest tests -q -m gpu ) > $OUT/pytest_pv2.log 2>&1; echo "pytest pv2 rc=$?"
  grep -a "passed\|failed" $OUT/pytest_pv2.log | tail -3 | tee -a $OUT/pv2.txt; grep -a "^FAILED" $OUT/pytest_pv2.log | head -40 | tee -a $OUT/pv2.txt
  ;;
s14)  # EXPERIMENT (experiments/r06_attn_longest_first.patch, _noorder = the committed library): packed rows: attention workgroups dispatched longest utterance first (order[] of gam_pack_index_kernel) vs batch order, same box, interleaved
  ( timeout 900 python -m pytest tests/test_hip_varlen.py -q -x ) > $OUT/varlen.log 2>&1; echo "varlen rc=$?"; tail -3 $OUT/varlen.log
  ( timeout 1200 python -m pytest tests -q -m gpu -x -k "ragged or live or fullsize or model_api" ) > $OUT/ragged.log 2>&1; echo "ragged rc=$?"; tail -4 $OUT/ragged.log
  C="--cpu-utts 0 --steps 12 --warmup 3 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power"
  for rep in 1 2 3; do
    for lib in libgigaam_hip_noorder.so libgigaam_hip.so; do
      for extra in "--config 2 --ragged" "--config 3 --ragged"; do
        ( GIGAAM_HIP_LIB=$R/gigaam_amd/$lib timeout 500 python bench.py $C $extra ) 2>> $OUT/ab.err | grep -a '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib $extra:', d['ms_per_step'], 'ms', d['value'], 'x', d.get('kernel_classes_ms_per_step'))" | tee -a $OUT/ab.txt
      done
    done
  done
  ;;
s15)  # EXPERIMENT (engine._decode_side_stream read GAM_SIDE_STREAM_PRIORITY for this run only): the overlapped RNN-T decode's side stream at normal vs high priority (a high-priority stream has its own hardware-queue pool)
  C="--cpu-utts 0 --steps 12 --warmup 3 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power --no-profile"
  for rep in 1 2 3; do
    for pr in 0 -1; do
      for extra in "--config 3" "--config 3 --ragged" "--config 4 --steps 4"; do
        ( GAM_SIDE_STREAM_PRIORITY=$pr timeout 500 python bench.py $C $extra ) 2>> $OUT/ab.err | grep -a '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('priority $pr $extra:', d['ms_per_step'], 'ms', d['value'], 'x')" | tee -a $OUT/ab.txt
      done
    done
  done
  ;;
s16)  # EXPERIMENT (needs profiles/experiments/r06_half_tiles.patch applied: it adds the kernels, gam_tune_sp_half, the test file and --half): half tiles (96 / 160 / 224 x 256): kernel tests, then the calibration sweep at packed-row and standard row counts
  ( timeout 900 python -m pytest tests/test_hip_half_tiles.py tests/test_hip_varlen.py -q -x ) > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/tests.log
  ( timeout 1500 python tools/smallm_sweep.py --calib --stages --half --rows=12017,5681,16064,8032,4016,2008,6432,10000,14000 ) > $OUT/sweep.txt 2> $OUT/sweep.err; echo "sweep rc=$?"; tail -3 $OUT/sweep.err
  cp gpurun_out/smallm_sweep.json $OUT/sweep.json 2>/dev/null
  ;;
s17)  # config 4 (packed rows + overlapped decode): run-to-run / box-to-box spread (final3 visit: 98.96 ms where final2 had 89.1), hardware-queue count, CUs held by the side decode
  C="--config 4 --cpu-utts 0 --steps 4 --warmup 2 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power --no-profile"
  run() { ( env "$@" timeout 500 python bench.py $C $X ) 2>> $OUT/ab.err | grep -a '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$* $X:', d['ms_per_step'], 'ms', d['value'], 'x')" | tee -a $OUT/ab.txt; }
  for rep in 1 2; do
    X="" run A=1
    X="" run GPU_MAX_HW_QUEUES=8
    X="--rnnt-side-cus 96" run A=1
    X="--rnnt-side-cus 224" run A=1
    X="--rnnt-overlap 0" run A=1
    X="--no-pack" run A=1
  done
  ;;
s20)  # auxiliary streams (decode side / collect / H2D copy) at normal vs high priority when the application took k streams from torch's pool first
  ( timeout 300 python tools/queue_probe.py ) > $OUT/queue.txt 2> $OUT/queue.err; grep -c "= 2.00" $OUT/queue.txt
  C="--cpu-utts 0 --steps 8 --warmup 2 --no-f32-leg --no-f16-leg --no-power --no-profile"
  for k in 0 1 2 3 4; do
    for pr in 0 -1; do
      for extra in "--config 3 --no-h2d-leg" "--config 4 --steps 4 --no-h2d-leg" "--config 5 --steps 3 --no-h2d-leg"; do
        ( GAM_AUX_STREAM_PRIORITY=$pr timeout 500 python bench.py $C $extra --pre-streams $k ) 2>> $OUT/ab.err | grep -a '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('pre-streams $k priority $pr $extra:', d['ms_per_step'], 'ms', d['value'], 'x')" | tee -a $OUT/ab.txt
      done
    done
  done
  ;;
s13)  # the one-switch reproducer, long: the same sources built WITH hipcc's SLP packing (libgigaam_hip_slp.so: 100+ v_pk_fma_f32 op_sel:[0,1,0]) and the
      # product library, same box, same script, no whole-CU claim
  repro slp     libgigaam_hip_slp.so GAM_RNNT_EXCLUSIVE=0 timeout 600 python tools/coresidency_repro.py 1,2 5000 gemm640
  repro product libgigaam_hip.so     GAM_RNNT_EXCLUSIVE=0 timeout 900 python tools/coresidency_repro.py 1,2 20000 gemm640
  repro slp2    libgigaam_hip_slp.so GAM_RNNT_EXCLUSIVE=0 timeout 600 python tools/coresidency_repro.py 1 5000 gemm640
  ;;
*) echo "unknown stage $STAGE"; exit 2;;
esac
