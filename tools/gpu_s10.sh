OUT=gpurun_out/r02_s10; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -a "passed\|failed\|Error\|error" $OUT/pytest.log | tail -8
python /tmp/c1.py default 2>/dev/null || { cat > /tmp/c1.py <<'PY'
import time, torch, sys
sys.path.insert(0,'.')
import gigaam_amd
from gigaam_amd import synth, workloads
ck = synth.make_checkpoint("v2_ctc", seed=0)
m = gigaam_amd.model_from_checkpoint(ck, "cuda:0")
eng = m.encoder.engine
wav, wlen = workloads.config1_clip(); wav, wlen = wav.cuda(), wlen.cuda()
def t(f, n=100):
    for _ in range(20): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
def core():
    enc, elen = eng.encode(*eng.frontend(wav, wlen)); return eng.ctc_greedy(enc, elen)
print(sys.argv[1], "config1 kernels only %.3f ms" % t(core))
PY
python /tmp/c1.py default; }
GAM_ROWSCALE=0 GAM_RANGE=0 python /tmp/c1.py both=0
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-utts 0 --no-f32-leg 2>/dev/null | grep -a "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('config2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_classes_ms_per_step'])"
timeout 300 python bench.py --config 5 --steps 3 --warmup 1 --cpu-utts 0 --no-profile 2>/dev/null | grep -a "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('config5', d['value'], d['ms_per_step'])"
