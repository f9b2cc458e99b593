for i in 1 2; do
GAM_BENCH_TRACE=1 timeout 300 python bench.py --config 5 --steps 3 --warmup 2 --cpu-utts 0 --no-profile 2>/tmp/err.log | grep -a "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('config5', d['value'], d['ms_per_step'])"
grep -a "trace" /tmp/err.log | tail -1 | cut -c1-900
done
timeout 300 python bench.py --config 4 --steps 3 --warmup 1 --cpu-utts 0 --no-profile 2>/dev/null | grep -a "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('config4', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --config 4 --steps 3 --warmup 1 --cpu-utts 0 --no-profile 2>/dev/null | grep -a "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('config4', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --config 3 --steps 10 --warmup 3 --cpu-utts 0 --no-f32-leg 2>/dev/null | grep -a "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('config3', d['value'], d['ms_per_step'])"
