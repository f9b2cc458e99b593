#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
GAM_GRAPH_DEBUG=1 timeout 300 python - <<'PY' > gpurun_out/graph.log 2>&1
import sys, time, torch
sys.path.insert(0, '.')
import gigaam_amd
from gigaam_amd import synth
ck = synth.make_checkpoint("v2_ctc", seed=0)
model = gigaam_amd.model_from_checkpoint(ck, "cuda:0")
eng = model.encoder.engine
wav, wlen = synth.synth_audio(1, 5.0, seed=1)
wav, wlen = wav.cuda(), wlen.cuda()
feat, flen = eng.frontend(wav, wlen)
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    enc, _ = eng.encode(feat, flen)
    torch.cuda.synchronize(); print("call", i, "%.3f ms" % ((time.perf_counter() - t0) * 1e3), float(enc.abs().sum()))
t0 = time.perf_counter()
for i in range(50): eng.encode(feat, flen)
torch.cuda.synchronize(); print("avg %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
del model, eng
import gc; gc.collect()
PY
cat gpurun_out/graph.log
