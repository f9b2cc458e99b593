#!/bin/bash
# Full measurement visit (one gpurun call, ~25 min): GPU tests with the measured-error report, smoke, the driver's bench command, a 1-rank
# torchrun launch, rocprofv3 kernel trace + PMC passes (separate runs, as the guide prescribes), all five BASELINE configurations incl. the
# ragged lines (packed and padded rows), the strong-scaling proxies, tools/scale8.sh at N = 1, the shared-device rehearsal of the multi-rank
# program, the co-residency reproducer on the product library and the packed-fp32 erratum canary.  Everything lands under gpurun_out/$TAG/;
# tools/collect_profiles.sh copies what is judged into profiles/.
#   gpurun --timeout 3000 -- 'bash tools/gpu_full.sh r06_final'
TAG=${1:-final}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export GAM_TEST_REPORT=$OUT/measured_errors.jsonl
rm -f $GAM_TEST_REPORT
( time timeout 1500 python -m pytest tests -q -m gpu ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
( timeout 300 python __graft_entry__.py --smoke ) > $OUT/smoke.log 2>&1; echo "smoke rc=$?"
( timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?"
( timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --cpu-utts 0 --no-f32-leg --no-h2d-leg --no-f16-leg ) > $OUT/bench_torchrun1.log 2>&1; echo "torchrun rc=$?"
cd /tmp
B="python $R/bench.py --steps 1 --warmup 1 --cpu-utts 0 --no-profile --no-f32-leg --no-h2d-leg --no-f16-leg"
( timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/pf_trace -o b -- python $R/bench.py --steps 5 --warmup 2 --cpu-utts 0 --no-f32-leg --no-h2d-leg --no-f16-leg ) > $OUT/pf_trace.log 2>&1
( timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pf_fetch -o b -- $B ) > $OUT/pf_fetch.log 2>&1
( timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pf_write -o b -- $B ) > $OUT/pf_write.log 2>&1
( timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pf_sq -o b -- $B ) > $OUT/pf_sq.log 2>&1
( timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/pf_trace_c3 -o b -- python $R/bench.py --config 3 --steps 3 --warmup 1 --cpu-utts 0 --no-f32-leg --no-h2d-leg --no-f16-leg ) > $OUT/pf_trace_c3.log 2>&1
( timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/pf_trace_b4 -o b -- python $R/bench.py --batch 4 --steps 10 --warmup 3 --cpu-utts 0 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power --no-profile ) > $OUT/pf_trace_b4.log 2>&1
( timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/pf_trace_c1 -o b -- python $R/bench.py --config 1 --steps 10 --warmup 3 --cpu-utts 0 --no-profile ) > $OUT/pf_trace_c1.log 2>&1
( timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/pf_trace_ragged -o b -- python $R/bench.py --config 2 --ragged --steps 5 --warmup 2 --cpu-utts 0 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power ) > $OUT/pf_trace_ragged.log 2>&1
cd $R
for n in trace fetch write sq trace_c3 trace_b4 trace_c1 trace_ragged; do
  DB=$(find $OUT/pf_$n -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB $OUT/${n}_summary.txt "rocprofv3 pass '$n' of bench.py (config 2, f16x3; trace_c3: config 3; trace_b4: --batch 4; trace_c1: config 1; trace_ragged: --config 2 --ragged, packed rows)" > /dev/null 2>&1
done
FD=$(find $OUT/pf_fetch -name "*.db" | head -1); WD=$(find $OUT/pf_write -name "*.db" | head -1)
[ -n "$FD" ] && [ -n "$WD" ] && python tools/pmc_traffic.py $FD $WD $OUT/pmc_traffic_f16x3.json > $OUT/pmc_traffic.log 2>&1
find $OUT -name "*.db" -delete
python tools/hbm_kernels.py $OUT/trace_summary.txt $OUT/fetch_summary.txt $OUT/write_summary.txt > $OUT/hbm_kernels.txt 2> $OUT/hbm_kernels.err
( timeout 900 python tools/bench_configs.py --only 1,3,4,5 --out $OUT/configs.jsonl ) > $OUT/configs.log 2>&1; echo "configs rc=$?"
# SURVEY 8d's second run of configs 2 / 3 (ragged lengths): packed rows (the default when the lengths are known on the host) and padded rows;
# the RNN-T decode in front of the encoder instead of beside it (--rnnt-overlap 0)
for extra in "--config 2 --ragged --no-f16-leg --no-h2d-leg" "--config 2 --ragged --no-pack --no-f16-leg --no-h2d-leg --cpu-utts 0" "--config 3 --ragged" "--config 3 --ragged --no-pack --cpu-utts 0" \
             "--config 3 --rnnt-overlap 0 --cpu-utts 0" "--config 4 --rnnt-overlap 0 --cpu-utts 0 --steps 4 --warmup 1" "--config 4 --no-pack --cpu-utts 0 --steps 4 --warmup 1"; do
  ( timeout 600 python bench.py $extra ) 2>> $OUT/extra.err | grep -a '^{' >> $OUT/extra_lines.jsonl; echo "bench $extra rc=$?"
done
( timeout 900 bash tools/scale8.sh $OUT/scale8 ) > $OUT/scale8.log 2>&1; echo "scale8 (N = 1 on this box) rc=$?"; tail -4 $OUT/scale8.log
bash tools/strong_proxy.sh $TAG > $OUT/strong_proxy.log 2>&1; cat $OUT/strong_proxy.log
# the multi-rank program on this 1-GPU box: plain --gpus 2 must refuse clearly; --oversubscribe runs it over gloo
( python bench.py --gpus 2 --steps 3 --warmup 1 ) > $OUT/rehearsal_refused.out 2> $OUT/rehearsal_refused.err; echo "plain --gpus 2 rc=$? (expected 1)"
for sc in weak strong; do
  ( timeout 300 python bench.py --gpus 2 --oversubscribe --scaling $sc --steps 5 --warmup 2 --no-profile --cpu-utts 0 ) 2> $OUT/rehearsal_$sc.err | grep -a '^{' > $OUT/rehearsal_gpus2_$sc.json; echo "rehearsal $sc rc=$?"
done
( timeout 300 python bench.py --gpus 2 --oversubscribe --config 5 --steps 1 --warmup 1 --no-profile --cpu-utts 0 ) 2> $OUT/rehearsal_c5.err | grep -a '^{' > $OUT/rehearsal_gpus2_config5.json; echo "rehearsal config5 x2 rc=$?"
( timeout 300 python bench.py --gpus 4 --oversubscribe --config 4 --utts-per-gpu 64 --steps 1 --warmup 1 --no-profile ) 2> $OUT/rehearsal_c4.err | grep -a '^{' > $OUT/rehearsal_gpus4_config4.json; echo "rehearsal config4 x4 rc=$?"
# r06: the co-residency reproducer on the PRODUCT library (no whole-CU claim) and the erratum canary (built here: ~20 s)
for c in "1,2,3,4 2000 gemm640"; do
  ( GAM_RNNT_EXCLUSIVE=0 timeout 600 python tools/coresidency_repro.py $c ) 2>> $OUT/repro.err | grep -a REPRO | tee -a $OUT/repro.txt
done
( hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/pkfma_rule.hip -o tools/libpkfma_rule.so && hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/l1_canary.hip -o tools/libl1_canary.so && timeout 300 python tools/pkfma_rule.py 2 ) > $OUT/pkfma_rule.txt 2> $OUT/pkfma_rule.err; echo "pkfma_rule rc=$?"
tail -3 $OUT/pytest_gpu.log; tail -1 $OUT/smoke.log
grep -a "^{" $OUT/bench.log | cut -c1-400
grep -a "^{" $OUT/bench_torchrun1.log | cut -c1-200
cat $OUT/configs.log | cut -c1-300
head -14 $OUT/trace_summary.txt | cut -c1-150
