#!/bin/bash
# r04 session 4: the planned stage count + the thread-per-frame CTC kernel: full GPU tests, then same-box A/B of two libraries
# at the small-grid points and the headline.   gpurun --timeout 1800 -- 'bash tools/r04_s4.sh r04_s4'   (A/B by GAM_SP_STAGES=2 vs planned)
TAG=${1:-r04_s4}; shift
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
export GAM_TEST_REPORT=$OUT/measured_errors.jsonl
( time timeout 1200 python -m pytest tests -q -x -m gpu ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -a "passed\|failed" $OUT/pytest_gpu.log | tail -2
C="--steps 30 --warmup 8 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power --cpu-utts 0 --no-profile"
one() {  # name env args
  local name=$1 e=$2; shift 2
  ( env $e timeout 300 python bench.py "$@" $C ) 2> $OUT/$name.err | grep -a '^{' > $OUT/$name.json
  python -c "
import json
try:
    d=json.loads(open('$OUT/$name.json').read()); print('$name', d['ms_per_step'], 'ms', d['value'], 'x')
except Exception as e: print('$name', 'FAILED', e)"
}
for rep in 1 2; do
  for v in ns2 plan; do
    E="X=1"; [[ $v == ns2 ]] && E="GAM_SP_STAGES=2"
    one c1_${v}_$rep $E --config 1
    one b2_${v}_$rep $E --batch 2
    one b4_${v}_$rep $E --batch 4
    one b8_${v}_$rep $E --batch 8
    one b16_${v}_$rep $E --batch 16
    one b32_${v}_$rep $E --batch 32
  done
done
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/pf_b4 -o b -- python $R/bench.py --batch 4 --steps 10 --warmup 3 --cpu-utts 0 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power --no-profile ) > $OUT/pf_b4.log 2>&1
DB=$(find $OUT/pf_b4 -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_summary.py $DB $OUT/trace_b4_summary.txt "rocprofv3 kernel trace, bench.py --batch 4 (r04: planned stages, thread-per-frame CTC kernel)" > /dev/null 2>&1
find $OUT -name "*.db" -delete
cut -c1-150 $OUT/trace_b4_summary.txt | head -24
