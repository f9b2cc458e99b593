#!/usr/bin/env python3
"""Kernel-only timing (HIP events around the launch) and fp64 error of the sp32 LDS-DMA GEMM
(gam_gemm_sp.h) against the 128x128 register-staged kernel, through gam_op_gemm."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigaam_amd import synth  # noqa: E402
from gigaam_amd.engine import HipEngine, build_config  # noqa: E402


def engine(sp, mt=None):
    os.environ["GAM_SP"] = str(sp)
    if mt:
        os.environ["GAM_SP_MT"] = str(mt)
    cfg = synth.model_cfg("v2_ctc")
    return HipEngine(build_config(cfg["preprocessor"], cfg["encoder"], None), {}, torch.device("cuda:0"))


def main():
    shapes = [(16064, 768, 768, 0), (16064, 3072, 768, 1), (16064, 768, 3072, 0), (16064, 2304, 768, 0),
              (16064, 1536, 768, 0), (16064, 768, 12288, 0), (4016, 768, 768, 0), (2500, 700, 96, 2)]
    if len(sys.argv) > 1 and sys.argv[1] == "quick":
        shapes = shapes[:3]
    elif len(sys.argv) > 1:      # "M,N,K,act;M,N,K,act;..."
        shapes = [tuple(int(v) for v in t.split(",")) for t in sys.argv[1].split(";") if t]
    engs = [("base", engine(0)), ("sp", engine(1))]
    if os.environ.get("GAM_SP_DBG"):
        engs = engs[1:]
    tlog = bool(os.environ.get("GAM_SP_TLOG"))    # instrumented library + GAM_SP_DBG=16: the launcher prints a timeline per call
    torch.manual_seed(0)
    for (m, n, k, act) in shapes:
        a = torch.randn(m, k, device="cuda")
        w = torch.randn(n, k, device="cuda") / k ** 0.5
        if os.environ.get("GAM_TEST_ZEROS"):      # data-dependent power: same kernel, all-zero operands
            a.zero_(); w.zero_()
        b = torch.randn(n, device="cuda")
        ref = a.double() @ w.double().t() + b.double()
        if act == 1:
            ref = ref * torch.sigmoid(ref)
        if act == 2:
            ref = ref.clamp_min(0)
        line = f"M={m:6d} N={n:5d} K={k:6d} act={act}"
        for name, e in engs:
            out = e.op_gemm(a, w, b, act)
            err = float((out.double() - ref).abs().max())
            if tlog:
                for _ in range(3):
                    e.op_gemm(a, w, b, act)
                torch.cuda.synchronize()
                print(f"==== {line} err {err:.1e}: the LAST [tlog] block above this line is the warm one", file=sys.stderr, flush=True)
                continue
            for _ in range(3):
                e.op_gemm(a, w, b, act)
            torch.cuda.synchronize()
            e.profile_enable(2)
            reps = 10
            for _ in range(reps):
                e.op_gemm(a, w, b, act)
            torch.cuda.synchronize()
            ms = e.profile_read()["gemm"]["ms"] / reps
            e.profile_enable(0)
            line += f" | {name}: {ms*1e3:8.1f} us {2.0*m*n*k/ms/1e9:6.1f} TF err {err:.1e}"
            if int(os.environ.get("GAM_SP_DBG", "0")) & 4:
                d = out[m - 1].cpu()
                for base in (0, 64):
                    for wv in (0, 5):
                        v = d[base + wv * 8: base + wv * 8 + 5].tolist()
                        if v[1] > 0:
                            line += (f"\n    tile@{base} wave{wv}: clk {v[0]:.0f} wall {v[1]:.0f} -> {v[0]/v[1]*100:.0f} MHz;"
                                     f" per k-tile: mm0 {v[2]/(k/32):.0f} bar {v[3]/(k/32):.0f} issue+rd+mm1 {v[4]/(k/32):.0f} clk")
        print(line, flush=True)


if __name__ == "__main__":
    main()
