#!/bin/bash
# One gpurun call: GPU parity tests (with the measured-error report), smoke, the headline bench, the other
# configurations, and a rocprofv3 kernel trace of the bench.  Everything lands under gpurun_out/$TAG/.
#   gpurun --timeout 1500 -- 'bash tools/gpu_session.sh r02_s1'
TAG=${1:-session}
WHAT=${2:-all}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export GAM_TEST_REPORT=$PWD/$OUT/measured_errors.jsonl
rm -f $GAM_TEST_REPORT
if [[ $WHAT == all || $WHAT == *test* ]]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
  tail -5 $OUT/pytest.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
fi
if [[ $WHAT == all || $WHAT == *bench* ]]; then
  timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_line.log 2> $OUT/bench_err.log; echo "bench rc=$?" | tee -a $OUT/summary.txt
  tail -c 3000 $OUT/bench_line.log
fi
if [[ $WHAT == all || $WHAT == *configs* ]]; then
  timeout 900 python tools/bench_configs.py --only 1,3,4,5 --out $OUT/configs.jsonl > $OUT/configs.log 2>&1; echo "configs rc=$?" | tee -a $OUT/summary.txt
  cat $OUT/configs.log
fi
if [[ $WHAT == all || $WHAT == *prof* ]]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-utts 0 --no-f32-leg --no-h2d-leg --no-f16-leg > $GRAFT_REPO_ROOT/$OUT/prof_bench.log 2>&1
  echo "rocprof rc=$?" | tee -a $GRAFT_REPO_ROOT/$OUT/summary.txt
  cd $GRAFT_REPO_ROOT
  DB=$(find $OUT/prof -name "*.db" | head -1)
  python tools/rocpd_summary.py $DB $OUT/trace_summary.txt "bench.py --steps 5 --warmup 2 (config 2, f16x3)" > /dev/null 2>&1 || true
  find $OUT/prof -name "*.db" -size +8M -delete
  head -30 $OUT/trace_summary.txt | cut -c1-200
fi
