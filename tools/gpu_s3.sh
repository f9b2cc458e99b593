OUT=gpurun_out/r02_s3; mkdir -p $OUT
export GAM_TEST_REPORT=$PWD/$OUT/measured_errors.jsonl; rm -f $GAM_TEST_REPORT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -a "passed\|failed\|Error\|error" $OUT/pytest.log | tail -8
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-utts 4 --no-f32-leg > $OUT/bench_line.log 2> $OUT/bench_err.log; echo "bench rc=$?"; grep -a "^{" $OUT/bench_line.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_classes_ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-utts 0 --no-f32-leg > $GRAFT_REPO_ROOT/$OUT/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $OUT/prof -name "*.db" | head -1)
python tools/rocpd_summary.py $DB $OUT/trace_summary.txt "bench.py --steps 5 --warmup 2 (config 2, f16x3)" > /dev/null 2>&1 || true
find $OUT/prof -name "*.db" -delete
head -16 $OUT/trace_summary.txt | cut -c1-170
