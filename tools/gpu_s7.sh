OUT=gpurun_out/r02_s7; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "v3 or encoder or ctc_bit or batching" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -a "passed\|failed\|Error\|error" $OUT/pytest.log | tail -8
for P in 0 1 0 1; do
  GAM_SP_PRIO=$P timeout 300 python bench.py --steps 20 --warmup 5 --cpu-utts 0 --no-f32-leg 2>/dev/null | grep -a "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('prio=$P', d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_classes_ms_per_step']['gemm'], d['kernel_classes_ms_per_step']['conv2'])"
done
timeout 300 python bench.py --config 4 --steps 3 --warmup 1 --cpu-utts 4 2>/dev/null | grep -a "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('config4', d['value'], d['ms_per_step'], d.get('kernel_classes_ms_per_step'), d.get('cpu_baseline',{}).get('gpu_ids_identical'))"
timeout 300 python bench.py --config 5 --steps 2 --warmup 1 --cpu-utts 0 2>$OUT/c5.err | grep -a "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('config5', d['value'], d['ms_per_step'], d['config']['workload'][:120])"
tail -3 $OUT/c5.err
