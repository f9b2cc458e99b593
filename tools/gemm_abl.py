import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigaam_amd import synth
from gigaam_amd.engine import HipEngine, build_config
cfg = synth.model_cfg("v2_ctc")
eng = HipEngine(build_config(cfg["preprocessor"], cfg["encoder"], None), {}, torch.device("cuda:0"))
eng.set_gemm_mode("f16x3")
for (m, n, k) in [(16064, 768, 3072), (16064, 3072, 768), (16064, 768, 12288)]:
    a = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") / k ** 0.5
    for _ in range(3): eng.op_gemm(a, w)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): eng.op_gemm(a, w)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"ablate={os.environ.get('GAM_ABLATE','0')} M={m} N={n} K={k} {dt*1e6:8.1f} us {2.0*m*n*k/dt/1e12:7.1f} TF", flush=True)
