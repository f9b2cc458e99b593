OUT=gpurun_out/r02_s4; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "encoder or ctc or rnnt or batched or fullsize or cluster" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -a "passed\|failed\|Error\|error" $OUT/pytest.log | tail -8
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-utts 0 --no-f32-leg > $OUT/bench_line.log 2> $OUT/bench_err.log; echo "bench rc=$?"; grep -a "^{" $OUT/bench_line.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_classes_ms_per_step'])"
for C in 8 5 4 2 1; do
  GAM_RNNT_CLUSTER=$C timeout 300 python bench.py --config 3 --steps 10 --warmup 3 --cpu-utts 0 --no-f32-leg 2>/dev/null | grep -a "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('config3 C=$C', d['value'], d['ms_per_step'], d['kernel_classes_ms_per_step']['decode'])"
done
for C in 8 4; do
  GAM_RNNT_CLUSTER=$C timeout 300 python bench.py --config 4 --steps 3 --warmup 1 --cpu-utts 4 2>/dev/null | grep -a "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('config4 C=$C', d['value'], d['ms_per_step'], d.get('kernel_classes_ms_per_step'), d.get('cpu_baseline',{}).get('gpu_ids_identical'))"
done
