for G in 1 0; do
echo "== GAM_GRAPH=$G"
GAM_GRAPH=$G GAM_BENCH_TRACE=1 timeout 300 python bench.py --config 5 --steps 3 --warmup 2 --cpu-utts 0 --no-profile 2>/tmp/err_$G.log | grep -a "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('config5', d['value'], d['ms_per_step'])"
grep -a "trace" /tmp/err_$G.log | tail -3 | cut -c1-1500
done
