import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gigaam_amd import synth
from gigaam_amd.engine import HipEngine, build_config
cfg = synth.model_cfg("v2_ctc")
eng = HipEngine(build_config(cfg["preprocessor"], cfg["encoder"], None), {}, torch.device("cuda:0"))
eng.set_gemm_mode("f16x3")
t=16
v = torch.zeros(1,t,48); v[0,torch.arange(16),torch.arange(16)] = 1.0   # ctx[q][key] = P[q][key]
torch.manual_seed(0)
for lo,hi in [(0,8),(8,16),(16,24),(24,32),(0,32),(32,36),(32,48)]:
    q = torch.zeros(1,t,48); k = torch.zeros(1,t,48)
    q[0,:,lo:hi] = torch.randn(t,hi-lo); k[0,:,lo:hi] = torch.randn(t,hi-lo)
    got = eng.op_attention(q,k,v).cpu().double()[0,:,:16]
    S = (q[0].double()@k[0].double().t())/48**0.5
    P = torch.softmax(S,-1)
    Sg = torch.log(got.clamp_min(1e-30)); Sg = Sg - Sg.mean(1,keepdim=True); Sr = S - S.mean(1,keepdim=True)
    print(f"d[{lo}:{hi}] max|P-Pref|={float((got-P).abs().max()):.3e} max|S-Sref|={float((Sg-Sr).abs().max()):.3e}")
    if lo==0 and hi==8:
        # which single d matches? test S computed with permuted d
        print("  S got row0:", [round(float(x),3) for x in Sg[0,:6]], "ref:", [round(float(x),3) for x in Sr[0,:6]])
