#!/bin/bash
# One GPU-box visit: A/B of the GEMM variants, bench, GPU tests, smoke, rocprof (+PMC passes).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 200 python tests/gpu_check.py --only gemm --out gpurun_out/check_gemm1.json ) > gpurun_out/check_gemm_nbuf1.log 2>&1
( GAM_GEMM_NBUF=2 timeout 200 python tests/gpu_check.py --only gemm --out gpurun_out/check_gemm2.json ) > gpurun_out/check_gemm_nbuf2.log 2>&1
( time timeout 400 python bench.py --steps 5 --warmup 2 ) > gpurun_out/bench.log 2>&1
( GAM_GEMM_NBUF=2 timeout 300 python bench.py --steps 5 --warmup 2 --cpu-utts 0 ) > gpurun_out/bench_nbuf2.log 2>&1
( time timeout 900 python -m pytest tests -q -m gpu ) > gpurun_out/pytest_gpu.log 2>&1
( time timeout 300 python __graft_entry__.py --smoke ) > gpurun_out/smoke.log 2>&1
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 3 --warmup 1 --cpu-utts 0 --no-profile ) > $R/gpurun_out/rocprof.log 2>&1
( timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof_pmc_fetch -o bench -- python $R/bench.py --steps 1 --warmup 1 --cpu-utts 0 --no-profile ) > $R/gpurun_out/rocprof_fetch.log 2>&1
( timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof_pmc_write -o bench -- python $R/bench.py --steps 1 --warmup 1 --cpu-utts 0 --no-profile ) > $R/gpurun_out/rocprof_write.log 2>&1
( timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $R/gpurun_out/prof_pmc_sq -o bench -- python $R/bench.py --steps 1 --warmup 1 --cpu-utts 0 --no-profile ) > $R/gpurun_out/rocprof_sq.log 2>&1
cd $R
grep gemm_time gpurun_out/check_gemm_nbuf1.log gpurun_out/check_gemm_nbuf2.log
tail -1 gpurun_out/bench.log | cut -c1-1500; tail -1 gpurun_out/bench_nbuf2.log | cut -c1-400
tail -5 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log
tail -2 gpurun_out/rocprof.log gpurun_out/rocprof_fetch.log gpurun_out/rocprof_write.log gpurun_out/rocprof_sq.log
ls gpurun_out/prof_*
