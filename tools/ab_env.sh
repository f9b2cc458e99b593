#!/bin/bash
# A/B of an environment switch on the headline step (same box, interleaved):  bash tools/ab_env.sh TAG VAR v1 v2 ... [-- extra bench args]
TAG=$1; VAR=$2; shift 2
VALS=(); while [[ $# -gt 0 && $1 != "--" ]]; do VALS+=("$1"); shift; done; [[ $1 == "--" ]] && shift
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=gpurun_out/$TAG; mkdir -p $OUT
C="--steps 20 --warmup 5 --no-f32-leg --no-h2d-leg --no-f16-leg --cpu-utts 0 $*"
for rep in 1 2; do
  for i in "${!VALS[@]}"; do
    v=${VALS[$i]}
    ( env $VAR=$v timeout 300 python bench.py $C ) > $OUT/${VAR}_${i}_$rep.json 2> $OUT/${VAR}_${i}_$rep.err
    echo "$VAR=$(basename $v) rep $rep: $(grep -a '^{' $OUT/${VAR}_${i}_$rep.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d.get("roofline",{}); print(d["ms_per_step"], d["value"], "frac", r.get("frac"), "sclk", r.get("avg_sclk_mhz_under_load"), d.get("kernel_classes_ms_per_step"), (d.get("board_power") or {}).get("avg_w"))' 2>&1)"
  done
done
