#!/bin/bash
# Final measurement visit of the round: tests, smoke, bench (both modes), rocprof trace + PMC passes.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -q -m gpu ) > gpurun_out/pytest_gpu.log 2>&1
( timeout 300 python __graft_entry__.py --smoke ) > gpurun_out/smoke.log 2>&1
( timeout 300 python bench.py --steps 10 --warmup 3 ) > gpurun_out/bench.log 2>&1
( timeout 300 python bench.py --steps 5 --warmup 2 --gemm f32 --cpu-utts 0 ) > gpurun_out/bench_f32.log 2>&1
( timeout 300 python bench.py --steps 5 --warmup 2 --model v2_rnnt --cpu-utts 4 ) > gpurun_out/bench_rnnt.log 2>&1
cd /tmp
B="python $R/bench.py --steps 1 --warmup 1 --cpu-utts 0 --no-profile"
( timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pf_trace -o b -- python $R/bench.py --steps 3 --warmup 1 --cpu-utts 0 --no-profile ) > $R/gpurun_out/pf_trace.log 2>&1
( timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pf_fetch -o b -- $B ) > $R/gpurun_out/pf_fetch.log 2>&1
( timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pf_write -o b -- $B ) > $R/gpurun_out/pf_write.log 2>&1
( timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pf_sq -o b -- $B ) > $R/gpurun_out/pf_sq.log 2>&1
cd $R
tail -3 gpurun_out/pytest_gpu.log; tail -1 gpurun_out/smoke.log
for f in bench bench_f32 bench_rnnt; do tail -1 gpurun_out/$f.log | cut -c1-200; done
ls gpurun_out/pf_*/
