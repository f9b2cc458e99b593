#!/bin/bash
# Strong-scaling proxies on ONE GPU: the per-rank workloads of the 2/4/8-GPU strong points of config 2 (batch 16 / 8 / 4),
# the single clip, and a kernel trace of the 4-utterance step.   gpurun --timeout 900 -- 'bash tools/r03_proxy.sh TAG'
TAG=${1:-r03_proxy}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
C="--steps 30 --warmup 8 --no-f32-leg --no-power --cpu-utts 0"
for b in 32 16 8 4; do
  ( timeout 300 python bench.py --batch $b $C ) > $OUT/b$b.json 2> $OUT/b$b.err; echo "batch $b rc=$?"
done
for m in 512 1024; do
  ( GAM_SP_MIN_M=$m timeout 300 python bench.py --batch 4 $C ) > $OUT/b4_spmin$m.json 2> $OUT/b4_spmin$m.err; echo "b4 spmin $m rc=$?"
  ( GAM_SP_MIN_M=$m timeout 300 python bench.py --batch 8 $C ) > $OUT/b8_spmin$m.json 2> $OUT/b8_spmin$m.err
done
( timeout 300 python bench.py --config 1 --steps 50 --warmup 10 --cpu-utts 0 ) > $OUT/c1.json 2> $OUT/c1.err; echo "config1 rc=$?"
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/pf_b4 -o b -- python $R/bench.py --batch 4 --steps 10 --warmup 3 --cpu-utts 0 --no-f32-leg --no-power --no-profile ) > $OUT/pf_b4.log 2>&1
( timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/pf_c1 -o b -- python $R/bench.py --config 1 --steps 10 --warmup 3 --cpu-utts 0 --no-profile ) > $OUT/pf_c1.log 2>&1
cd $R
for n in b4 c1; do
  DB=$(find $OUT/pf_$n -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB $OUT/${n}_trace_summary.txt "rocprofv3 kernel trace of bench.py ($n)" > /dev/null 2>&1
done
find $OUT -name "*.db" -delete
for f in $OUT/b*.json $OUT/c1.json; do echo "$(basename $f): $(grep -a '^{' $f | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d.get("kernel_classes_ms_per_step"))' 2>&1)"; done
head -25 $OUT/b4_trace_summary.txt | cut -c1-160
