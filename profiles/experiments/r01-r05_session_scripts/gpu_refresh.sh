#!/bin/bash
# Short evidence refresh (bench line, kernel trace, config 1 / batch 4 points, PMC traffic) -- tools/gpu_final.sh is the full visit.
TAG=${1:-refresh}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
( timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?"
cd /tmp
( timeout 60 rocprofv3 --kernel-trace --stats -d $OUT/pf_trace -o b -- python $R/bench.py --steps 5 --warmup 2 --cpu-utts 0 --no-f32-leg --no-h2d-leg --no-f16-leg ) > $OUT/pf_trace.log 2>&1
cd $R
DB=$(find $OUT/pf_trace -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB $OUT/trace_summary.txt "rocprofv3 pass 'trace' of bench.py (config 2, f16x3)" > /dev/null 2>&1
( timeout 40 python bench.py --config 1 --steps 50 --warmup 10 --cpu-utts 0 --no-profile ) 2> $OUT/c1.err | grep -a '^{' > $OUT/c1.json
( timeout 40 python bench.py --batch 4 --steps 30 --warmup 8 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power --cpu-utts 0 --no-profile ) 2> $OUT/b4.err | grep -a '^{' > $OUT/b4.json
cd /tmp
B="python $R/bench.py --steps 1 --warmup 1 --cpu-utts 0 --no-profile --no-f32-leg --no-h2d-leg --no-f16-leg"
( timeout 50 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pf_fetch -o b -- $B ) > $OUT/pf_fetch.log 2>&1
( timeout 50 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pf_write -o b -- $B ) > $OUT/pf_write.log 2>&1
cd $R
FD=$(find $OUT/pf_fetch -name "*.db" | head -1); WD=$(find $OUT/pf_write -name "*.db" | head -1)
[ -n "$FD" ] && [ -n "$WD" ] && python tools/pmc_traffic.py $FD $WD $OUT/pmc_traffic_f16x3.json > $OUT/pmc_traffic.log 2>&1
for n in fetch write; do DB=$(find $OUT/pf_$n -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py $DB $OUT/${n}_summary.txt "rocprofv3 pass '$n' of bench.py (config 2, f16x3)" > /dev/null 2>&1; done
find $OUT -name "*.db" -delete
grep -a '^{' $OUT/bench.log | cut -c1-300; head -8 $OUT/trace_summary.txt | cut -c1-150
python -c "
import json
for n in ('c1','b4'):
    d=json.loads(open('$OUT/'+n+'.json').read()); print(n, d['ms_per_step'], d['value'])"
