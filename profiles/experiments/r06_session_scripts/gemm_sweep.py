#!/usr/bin/env python3
"""Microbenchmark of gam_gemm_f32_kernel through the C ABI (gam_op_gemm): K / N / M sweeps
at the bench's token count, to separate per-launch fixed cost from steady-state MFMA rate."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigaam_amd import synth  # noqa: E402
from gigaam_amd.engine import HipEngine, build_config  # noqa: E402


def main():
    cfg = synth.model_cfg("v2_ctc")
    eng = HipEngine(build_config(cfg["preprocessor"], cfg["encoder"], None), {}, torch.device("cuda:0"))
    eng.set_gemm_mode(sys.argv[1] if len(sys.argv) > 1 else "f16x3")
    print("mode", eng.gemm_mode)
    res = []
    shapes = []
    for k in (128, 256, 768, 1536, 3072, 6144, 12288):
        shapes.append((16064, 768, k, 0))
    for k in (256, 768, 1536):
        shapes.append((16064, 3072, k, 0))
    shapes += [(16064, 3072, 768, 1), (16064, 1536, 768, 0), (16064, 2304, 768, 0), (16064, 320, 768, 0), (16064, 34, 768, 0)]
    for m in (126, 1004, 2008, 4016, 8032, 32128):
        shapes.append((m, 768, 768, 0))
    for (m, n, k, act) in shapes:
        a = torch.randn(m, k, device="cuda")
        w = torch.randn(n, k, device="cuda") / k ** 0.5
        b = torch.randn(n, device="cuda")
        if m * n <= 16064 * 768:
            ref = (a.double() @ w.double().t() + b.double())
            if act == 1:
                ref = ref * torch.sigmoid(ref)
            err = float((eng.op_gemm(a, w, b, act).double() - ref).abs().max())
        else:
            err = -1.0
        for _ in range(3):
            eng.op_gemm(a, w, b, act)
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.op_gemm(a, w, b, act)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        tf = 2.0 * m * n * k / dt / 1e12
        res.append({"M": m, "N": n, "K": k, "act": act, "us": dt * 1e6, "tflops": tf})
        print(f"M={m:6d} N={n:5d} K={k:6d} act={act}  {dt*1e6:9.1f} us  {tf:7.1f} TF  max_err={err:.2e}", flush=True)
    out = os.path.join(ROOT, "gpurun_out", "gemm_sweep.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
