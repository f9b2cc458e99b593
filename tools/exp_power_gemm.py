#!/usr/bin/env python3
"""Board power and shader clock (amdgpu hwmon, bench.PowerSampler) while one large split-fp16 GEMM runs back to
back for a few seconds, on zero operands and on N(0,1) operands: the direct form of DESIGN's "the GEMM main loop is
power-limited" (same kernel, same instruction stream; only the operands' bit activity differs).
gam_op_gemm converts A and W to the split layout on every call (~10 % of each iteration at this shape)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gigaam_amd import synth  # noqa: E402
from gigaam_amd.engine import HipEngine, build_config  # noqa: E402


def main():
    cfg = synth.model_cfg("v2_ctc")
    eng = HipEngine(build_config(cfg["preprocessor"], cfg["encoder"], None), {}, torch.device("cuda:0"))
    eng.set_gemm_mode("f16x3")
    m, n, k = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (16064, 3072, 3072)
    out = []
    for name in ("zeros", "randn", "zeros", "randn"):
        a = torch.zeros(m, k, device="cuda") if name == "zeros" else torch.randn(m, k, device="cuda")
        w = torch.zeros(n, k, device="cuda") if name == "zeros" else torch.randn(n, k, device="cuda") / k ** 0.5
        for _ in range(5):
            eng.op_gemm(a, w)
        torch.cuda.synchronize()
        reps = 0
        with bench.PowerSampler(0) as ps:
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 3.0:
                for _ in range(50):
                    eng.op_gemm(a, w)
                torch.cuda.synchronize()
                reps += 50
            dt = time.perf_counter() - t0
        rec = {"operands": name, "M": m, "N": n, "K": k, "us_per_call": round(dt / reps * 1e6, 1),
               "algorithmic_tflops": round(2.0 * m * n * k * reps / dt / 1e12, 1), **ps.summary()}
        out.append(rec)
        print(json.dumps(rec), flush=True)
    path = os.path.join(ROOT, "gpurun_out", f"power_gemm_{m}x{n}x{k}.jsonl")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        for r in out:
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
