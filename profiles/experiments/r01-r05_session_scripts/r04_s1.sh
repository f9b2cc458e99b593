#!/bin/bash
# r04 session 1 (diagnostics): the small-grid questions VERDICT r3 left open, on ONE box.
#   gpurun --timeout 900 -- 'bash tools/r04_s1.sh r04_s1'
#  (a) where do the 23 extra __amd_rocclr_copyBuffer dispatches per hipGraph replay come from (runtime knobs)
#  (b) does the graph pay above 2048 rows (GAM_GRAPH_MAX_ROWS)
#  (c) one batch as two part-batches on two streams at the strong-scaling points
TAG=${1:-r04_s1}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
C="--steps 30 --warmup 8 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power --cpu-utts 0 --no-profile"
one() {  # name env... -- args
  local name=$1; shift
  local envs=(); while [[ $# -gt 0 && $1 != "--" ]]; do envs+=("$1"); shift; done; shift
  ( env "${envs[@]}" timeout 300 python bench.py "$@" $C ) 2> $OUT/$name.err | grep -a '^{' > $OUT/$name.json
  python - "$OUT/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    print(sys.argv[2], d["ms_per_step"], "ms", d["value"], "x")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
  one b4_default_$rep X=1 -- --batch 4
  one b4_nocapture_$rep DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 -- --batch 4
done
one b4_batch1024 DEBUG_HIP_GRAPH_BATCH_SIZE=1024 -- --batch 4
one b4_nograph GAM_GRAPH=0 -- --batch 4
one c1_default X=1 -- --config 1
one c1_nocapture DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 -- --config 1
one b8_default X=1 -- --batch 8
one b8_graph GAM_GRAPH_MAX_ROWS=100000 -- --batch 8
one b16_default X=1 -- --batch 16
one b16_graph GAM_GRAPH_MAX_ROWS=100000 -- --batch 16
one b32_default X=1 -- --batch 32
# kernel traces: copyBuffer count per step under the knobs
cd /tmp
for v in default nocapture; do
  E="X=1"; [[ $v == nocapture ]] && E="DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
  ( env $E timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/pf_b4_$v -o b -- python $R/bench.py --batch 4 --steps 10 --warmup 3 --cpu-utts 0 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power --no-profile ) > $OUT/pf_b4_$v.log 2>&1
  DB=$(find $OUT/pf_b4_$v -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_summary.py $DB $OUT/trace_b4_${v}_summary.txt "rocprofv3 kernel trace, bench.py --batch 4 ($v)" > /dev/null 2>&1
  grep -a "copyBuffer\|fillBuffer" $OUT/trace_b4_${v}_summary.txt | cut -c1-160
done
find $OUT -name "*.db" -delete
cd $R
for b in 4 8; do
  ( timeout 300 python tools/exp_two_streams.py --batch $b --parts 1,2,1,2,4 --reps 30 ) > $OUT/two_streams_b$b.log 2>&1
  cat $OUT/two_streams_b$b.log | grep -a stream
done
