cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/ab_qkv; mkdir -p $OUT
( timeout 300 python -m pytest tests -q -m gpu -x ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -aE "passed|failed|Error" $OUT/pytest.log | tail -3
one() { name=$1; lib=$2; shift 2; ( GIGAAM_HIP_LIB=$lib timeout 120 python bench.py "$@" --cpu-utts 0 ) 2> $OUT/$name.err | grep -a '^{' > $OUT/$name.json; python -c "
import json,sys; d=json.loads(open('$OUT/$name.json').read()); print('$name', d['ms_per_step'], d['value'], d.get('kernel_classes_ms_per_step'), (d.get('roofline') or {}).get('frac'))"; }
for i in 0 1; do
  lib=gigaam_amd/libgigaam_hip_old.so; [ $i = 1 ] && lib=gigaam_amd/libgigaam_hip.so
  one c1_$i $lib --config 1 --steps 50 --warmup 10 --no-profile
  one b4_$i $lib --batch 4 --steps 30 --warmup 8 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power --no-profile
  one head_$i $lib --steps 20 --warmup 5 --no-f32-leg --no-h2d-leg --no-f16-leg --no-power
done
