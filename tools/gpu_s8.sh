OUT=gpurun_out/r02_s8; mkdir -p $OUT
for G in 1 0; do
GAM_GRAPH=$G GAM_GRAPH_DEBUG=1 timeout 300 python bench.py --config 1 --steps 50 --warmup 10 --no-profile --cpu-utts 0 2>$OUT/c1_$G.err | grep -a "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('config1 graph=$G', d['value'], d['ms_per_step'])"
grep -a "graph replays" $OUT/c1_$G.err | tail -1
done
timeout 300 python bench.py --config 1 --steps 50 --warmup 10 --cpu-utts 0 2>/dev/null | grep -a "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('config1 profiled', d['value'], d['ms_per_step'], d['kernel_classes_ms_per_step'])"
timeout 300 python bench.py --config 5 --steps 2 --warmup 1 --cpu-utts 0 2>/dev/null | grep -a "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('config5 profiled', d['value'], d['ms_per_step'], d['kernel_classes_ms_per_step'])"
timeout 300 python bench.py --config 5 --steps 3 --warmup 1 --cpu-utts 0 --no-profile 2>/dev/null | grep -a "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('config5', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --config 5 --fr-batch 32 --steps 3 --warmup 1 --cpu-utts 0 --no-profile 2>/dev/null | grep -a "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('config5 fr32', d['value'], d['ms_per_step'])"
python - <<'PY'
import time, torch, sys
sys.path.insert(0,'.')
import gigaam_amd
from gigaam_amd import synth, workloads
ck = synth.make_checkpoint("v2_ctc", seed=0)
m = gigaam_amd.model_from_checkpoint(ck, "cuda:0")
eng = m.encoder.engine
wav, wlen = workloads.config1_clip(); wav, wlen = wav.cuda(), wlen.cuda()
def t(f, n=50):
    for _ in range(10): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
def core():
    enc, elen = eng.encode(*eng.frontend(wav, wlen)); return eng.ctc_greedy(enc, elen)
print("config1 kernels only %.3f ms" % t(core))
print("  + range_flag %.3f" % t(lambda: (core(), eng.range_flag())))
def full():
    ids, fr, c = core(); eng.range_flag(); n=c.cpu().tolist(); w=max(n); return ids[:, :w].cpu(), fr[:, :w].cpu()
print("  + range_flag + D2H %.3f" % t(full))
PY
