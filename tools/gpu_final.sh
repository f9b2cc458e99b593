#!/bin/bash
# Final measurement visit of the round: tests, smoke, bench (both modes + a 1-rank torchrun launch of the
# distributed path), rocprof trace + PMC passes, all five BASELINE configurations.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -q -m gpu ) > gpurun_out/pytest_gpu.log 2>&1
( timeout 300 python __graft_entry__.py --smoke ) > gpurun_out/smoke.log 2>&1
( timeout 300 python bench.py --steps 10 --warmup 3 ) > gpurun_out/bench.log 2>&1
( timeout 300 python bench.py --steps 5 --warmup 2 --gemm f32 --cpu-utts 0 ) > gpurun_out/bench_f32.log 2>&1
( timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --cpu-utts 0 ) > gpurun_out/bench_torchrun1.log 2>&1
cd /tmp
B="python $R/bench.py --steps 1 --warmup 1 --cpu-utts 0 --no-profile"
( timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pf_trace -o b -- python $R/bench.py --steps 3 --warmup 1 --cpu-utts 0 --no-profile ) > $R/gpurun_out/pf_trace.log 2>&1
( timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pf_fetch -o b -- $B ) > $R/gpurun_out/pf_fetch.log 2>&1
( timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pf_write -o b -- $B ) > $R/gpurun_out/pf_write.log 2>&1
( timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pf_sq -o b -- $B ) > $R/gpurun_out/pf_sq.log 2>&1
cd $R
( timeout 600 python tools/bench_configs.py --only 1,3,4,5 ) > gpurun_out/configs.log 2>&1
tail -3 gpurun_out/pytest_gpu.log; tail -1 gpurun_out/smoke.log
for f in bench bench_f32 bench_torchrun1; do tail -1 gpurun_out/$f.log | cut -c1-200; done
tail -6 gpurun_out/configs.log | cut -c1-260
ls gpurun_out/pf_*/
