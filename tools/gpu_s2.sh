OUT=gpurun_out/r02_s2; mkdir -p $OUT
export GAM_TEST_REPORT=$PWD/$OUT/measured_errors.jsonl; rm -f $GAM_TEST_REPORT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -a "passed\|failed\|Error\|error" $OUT/pytest.log | tail -8
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-utts 4 --no-f32-leg > $OUT/bench_line.log 2> $OUT/bench_err.log; echo "bench rc=$?"; grep -a "^{" $OUT/bench_line.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_classes_ms_per_step'], d['cpu_baseline'])"
timeout 600 python tools/bench_configs.py --only 3,4 --out $OUT/configs.jsonl > $OUT/configs.log 2>&1; cat $OUT/configs.log
GAM_RNNT_CLUSTER=0 timeout 600 python tools/bench_configs.py --only 3,4 --out $OUT/configs_nocluster.jsonl > $OUT/configs_nocluster.log 2>&1; cat $OUT/configs_nocluster.log
timeout 300 python tools/exp_two_streams.py > $OUT/two_streams.log 2>&1; cat $OUT/two_streams.log
