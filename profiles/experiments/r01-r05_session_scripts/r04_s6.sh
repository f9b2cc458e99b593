#!/bin/bash
# r04 session 6: the driver's command with the new legs, the other configurations, the published shapes, the strong proxies.
TAG=${1:-r04_s6}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
( timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?"
grep -a '^{' $OUT/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d.get(k) for k in ('value','ms_per_step','h2d_ms','gather_path')}); print(d.get('h2d')); print(d['roofline']['frac'], d['cpu_baseline'].get('value'), d['cpu_baseline'].get('gpu_ids_identical'), d['cpu_baseline'].get('reference_rtfx_build_container'))"
( timeout 900 python tools/bench_configs.py --only 1,3,4,5 --out $OUT/configs.jsonl ) > $OUT/configs.log 2>&1; echo "configs rc=$?"; cut -c1-300 $OUT/configs.log
python -c "
import json
for ln in open('$OUT/configs.jsonl'):
    d=json.loads(ln); r=d.get('roofline') or {}
    print(d['config']['baseline_config'], d['value'], d['ms_per_step'], 'frac', r.get('frac'), (d['config'].get('dealing') or {}).get('dealing_bound_speedup'))"
( timeout 600 python tools/published_shapes.py --out $OUT/published_shapes.jsonl ) > $OUT/published.log 2>&1; echo "published rc=$?"; cat $OUT/published.log | grep -a '^{'
bash tools/strong_proxy.sh $TAG > $OUT/strong_proxy.log 2>&1; cat $OUT/strong_proxy.log | cut -c1-250
