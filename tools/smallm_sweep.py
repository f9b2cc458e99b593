#!/usr/bin/env python3
"""Per-shape kernel time (HIP events around gemm(): the GEMM launch + its split-K reduce) of the Conformer layer's GEMM
shapes at the row counts of the strong-scaling points (2 / 4 / 8 / 16 utterances of 20 s per GPU, one 5 s clip):
  base      the 128x128 register-staged kernels + split-K (what runs below GAM_SP_MIN_M rows)
  sp        the LDS-DMA sp32 kernel as planned by gam_gemm_sp_plan (tile shape + split-K from its time model)
  sweep     with --calib: EVERY (MT, NW, S) of the sp kernel through the gam_tune_sp hook -- the data the plan's model is fitted to
    python tools/smallm_sweep.py [--calib] > profiles/r03_smallm_sweep.txt"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigaam_amd import synth  # noqa: E402
from gigaam_amd.engine import HipEngine, build_config  # noqa: E402


def engine(sp_min):
    os.environ["GAM_SP_MIN_M"] = str(sp_min)
    cfg = synth.model_cfg("v2_ctc")
    return HipEngine(build_config(cfg["preprocessor"], cfg["encoder"], None), {}, torch.device("cuda:0"))


def timed(e, a, w, b, act, reps=10):
    for _ in range(3):
        e.op_gemm(a, w, b, act)
    torch.cuda.synchronize()
    e.profile_enable(2)
    for _ in range(reps):
        e.op_gemm(a, w, b, act)
    torch.cuda.synchronize()
    ms = e.profile_read()["gemm"]["ms"]
    e.profile_enable(0)
    return ms / reps * 1e3


def main():
    calib = "--calib" in sys.argv
    stages = "--stages" in sys.argv       # r04: every configuration also with three LDS stages where the tile has that build
    rows_arg = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--rows=")), None)
    m_list = [int(v) for v in rows_arg.split(",")] if rows_arg else [126, 1004, 2008, 4016, 8032, 16064]
    base, sp = engine(1 << 30), engine(1)
    lib = sp.lib
    torch.manual_seed(0)
    rows = []
    for m in m_list:
        for (n, k, act) in ((3072, 768, 1), (768, 3072, 0), (1536, 768, 0), (768, 768, 0), (2304, 768, 0), (768, 12288, 0)):
            a = torch.randn(m, k, device="cuda")
            w = torch.randn(n, k, device="cuda") / k ** 0.5
            b = torch.randn(n, device="cuda")
            ref = a.double() @ w.double().t() + b.double()
            if act == 1:
                ref = ref * torch.sigmoid(ref)
            lib.gam_tune_sp(0, 0, 0)
            lib.gam_tune_sp_stages(0)
            err = float((sp.op_gemm(a, w, b, act).double() - ref).abs().max() / ref.abs().max())
            rec = {"M": m, "N": n, "K": k, "base": timed(base, a, w, b, act), "sp": timed(sp, a, w, b, act), "rel_err": err}
            line = (f"M={m:5d} N={n:5d} K={k:6d} act={act}  ideal@360TF {2.0*m*n*k/360e6:6.1f} us | base {rec['base']:7.1f} | "
                    f"sp(plan) {rec['sp']:7.1f} us  rel.err {err:.1e}")
            if calib:
                allc = {}
                for (mt, nw) in ((2, 4), (3, 4), (4, 4), (2, 2), (3, 2)):
                    for s_ in (1, 2, 3, 4, 6, 8):
                        nk = k // 32
                        if s_ > 1 and (nk % s_ or nk // s_ < 4):
                            continue
                        lib.gam_tune_sp(mt, nw, s_)
                        has3 = (nw == 2 and mt in (2, 3)) or (nw == 4 and mt == 2)
                        for ns in ((2, 3) if (stages and has3) else (2,)):
                            if ns == 3 and s_ > 1 and nk // s_ < 4:
                                continue
                            lib.gam_tune_sp_stages(ns)
                            out = sp.op_gemm(a, w, b, act)
                            e2 = float((out.double() - ref).abs().max() / ref.abs().max())
                            assert e2 < 3e-5, (m, n, k, mt, nw, s_, ns, e2)
                            allc[f"{mt}x{nw}/S{s_}" + ("/n3" if ns == 3 else "")] = round(timed(sp, a, w, b, act, reps=6), 1)
                lib.gam_tune_sp(0, 0, 0)
                lib.gam_tune_sp_stages(0)
                rec["configs"] = allc
                top = sorted(allc.items(), key=lambda kv: kv[1])[:8]
                line += " | best: " + ", ".join(f"{c} {t}" for c, t in top)
            rows.append(rec)
            print(line, flush=True)
    out = os.path.join(ROOT, "gpurun_out", "smallm_sweep.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rows, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
