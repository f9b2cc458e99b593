OUT=gpurun_out/r02_s5; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "rnnt or cluster" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -a "passed\|failed\|Error\|error" $OUT/pytest.log | tail -8
for C in 7 5; do
  GAM_RNNT_CLUSTER=$C timeout 300 python bench.py --config 3 --steps 10 --warmup 3 --cpu-utts 4 --no-f32-leg 2>/dev/null | grep -a "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('config3 C=$C', d['value'], d['ms_per_step'], d['kernel_classes_ms_per_step']['decode'], d['cpu_baseline']['gpu_ids_identical'])"
done
for C in 7; do
  GAM_RNNT_CLUSTER=$C timeout 300 python bench.py --config 4 --steps 3 --warmup 1 --cpu-utts 4 2>/dev/null | grep -a "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('config4 C=$C', d['value'], d['ms_per_step'], d.get('kernel_classes_ms_per_step'), d.get('cpu_baseline',{}).get('gpu_ids_identical'))"
done
