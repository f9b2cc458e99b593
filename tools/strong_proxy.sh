#!/bin/bash
# Strong-scaling proxies: what ONE rank of a 1/2/4/8/16-GPU strong split of config 2 runs (32/16/8/4/2 utterances of 20 s), and
# the single clip, on ONE GPU.  The step time comes from a run WITHOUT per-launch HIP events (--no-profile: at 4 x 20 s the
# ~300 event pairs of the roofline leg cost ~1 ms of an 8 ms step); the per-class breakdown from a second, profiled run.
#   gpurun --timeout 900 -- 'bash tools/strong_proxy.sh TAG'   ->  gpurun_out/TAG/strong_proxy.jsonl
TAG=${1:-strong_proxy}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/$TAG; mkdir -p $OUT
C="--steps 30 --warmup 8 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power --cpu-utts 0"
rm -f $OUT/strong_proxy.jsonl
for b in 32 16 8 4 2; do
  ( timeout 300 python bench.py --batch $b $C --no-profile ) 2> $OUT/b$b.err | grep -a '^{' > $OUT/b$b.json
  ( timeout 300 python bench.py --batch $b $C ) 2>> $OUT/b$b.err | grep -a '^{' > $OUT/b${b}_profiled.json
  python - $OUT/b$b.json $OUT/b${b}_profiled.json $OUT/strong_proxy.jsonl <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read()); p = json.loads(open(sys.argv[2]).read())
rec = {"utterances_per_rank": d["config"]["global_batch"], "ms_per_step": d["ms_per_step"], "rtfx": d["value"],
       "ms_per_step_with_per_launch_events": p["ms_per_step"], "kernel_classes_ms_per_step": p.get("kernel_classes_ms_per_step"),
       "command": "bench.py --batch N --steps 30 --warmup 8 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power --cpu-utts 0 --no-profile"}
open(sys.argv[3], "a").write(json.dumps(rec) + "\n")
print("batch", rec["utterances_per_rank"], rec["ms_per_step"], "ms (with events:", rec["ms_per_step_with_per_launch_events"], ")", rec["kernel_classes_ms_per_step"])
PY
done
( timeout 300 python bench.py --config 1 --steps 50 --warmup 10 --cpu-utts 0 --no-profile ) 2> $OUT/c1.err | grep -a '^{' > $OUT/c1.json
python -c "import json,sys; d=json.loads(open('$OUT/c1.json').read()); print('config1:', d['ms_per_step'], d['value'])"
