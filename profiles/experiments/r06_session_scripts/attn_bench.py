#!/usr/bin/env python3
"""Kernel-only time (HIP events) of the fused attention through gam_op_attention at the bench shape."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gigaam_amd import synth
from gigaam_amd.engine import HipEngine, build_config
cfg = synth.model_cfg("v2_ctc")
eng = HipEngine(build_config(cfg["preprocessor"], cfg["encoder"], None), {}, torch.device("cuda:0"))
b, t, h = 32, 502, 16
g = torch.Generator().manual_seed(0)
q, k, v = (torch.randn(b, t, h * 48, generator=g).cuda() * s for s in (2.0, 2.0, 1.0))
lens = torch.full((b,), 501, dtype=torch.int32).cuda()
ref = None
for _ in range(3): out = eng.op_attention(q, k, v, lens)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): eng.op_attention(q, k, v, lens)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
qh, kh, vh = (x[:2].double().view(2, t, h, 48).transpose(1, 2) for x in (q, k, v))
sc = qh @ kh.transpose(-1, -2) / 48 ** 0.5
sc = sc.masked_fill((torch.arange(t, device="cuda")[None, :] >= 501)[None, None], float("-inf"))
want = (torch.softmax(sc, -1) @ vh).transpose(1, 2).reshape(2, t, h * 48)
print(os.environ.get("GIGAAM_HIP_LIB", "default"), f"{ms*1e3:.1f} us", "err %.1e" % float((out[:2, :501].double() - want[:, :501]).abs().max()))
