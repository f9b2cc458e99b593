import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from gigaam_amd import synth
from gigaam_amd.engine import HipEngine, build_config
m, n, k = (int(x) for x in sys.argv[1:4])
cfg = synth.model_cfg("v2_ctc")
eng = HipEngine(build_config(cfg["preprocessor"], cfg["encoder"], None), {}, torch.device("cuda:0"))
eng.set_gemm_mode(sys.argv[4] if len(sys.argv) > 4 else "f16x3")
a = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") / k ** 0.5
for _ in range(4): eng.op_gemm(a, w)
torch.cuda.synchronize()
