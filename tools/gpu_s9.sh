cat > /tmp/c1.py <<'PY'
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
python /tmp/c1.py default
GAM_ROWSCALE=0 python /tmp/c1.py rowscale=0
GAM_RANGE=0 python /tmp/c1.py range=0
GAM_ROWSCALE=0 GAM_RANGE=0 python /tmp/c1.py both=0
GAM_ROWSCALE=0 GAM_RANGE=0 GAM_GRAPH=0 python /tmp/c1.py both=0,graph=0
GAM_SPLITK=0 python /tmp/c1.py splitk=0
