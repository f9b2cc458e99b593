#!/usr/bin/env python3
"""The reference's only published latency figures are "Full Encoder Inference" at bs x seq_len = 1 x 10 s, 8 x 20 s and
128 x 30 s (/root/reference/evaluation.md:61-67: 10.1 / 15.8 / 324.5 ms on an unnamed CUDA GPU, fp16 autocast).  This
tool times the SAME shapes here -- encoder only (gam_encode on log-mel features already in HBM) and the whole path
(log-mel + encoder + CTC greedy + ids on the host) -- in both arithmetic modes, one JSON line per (shape, mode):

    python tools/published_shapes.py [--out gpurun_out/published_shapes.jsonl] [--reps 20]

Context, not a same-node comparison: different silicon, and the reference figure is fp16 while both legs here are
fp32-accurate.  Also answers "does anything fall off a cliff above batch 32" (128 x 30 s: T' = 751, 96 128 token rows)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gigaam_amd  # noqa: E402
from gigaam_amd import synth  # noqa: E402
from gigaam_amd.engine import HipEngine  # noqa: E402

PUBLISHED_MS = {(1, 10): 10.06, (8, 20): 15.90, (128, 30): 324.48}     # SDPA column, evaluation.md:65-67
SHAPES = [(1, 10), (8, 20), (128, 30)]


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "published_shapes.jsonl"))
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--model", default="v2_ctc")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ck = synth.make_checkpoint(args.model, seed=0)
    model = gigaam_amd.model_from_checkpoint(ck, dev)
    eng = model.encoder.engine
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        for b, sec in SHAPES:
            wav, wlen = synth.synth_audio(b, float(sec), seed=1000)
            wav, wlen = wav.to(dev), wlen.to(dev)
            for mode in ("f16x3", "f32", "f16"):
                eng.set_gemm_mode(mode)
                feat, flen = eng.frontend(wav, wlen)
                reps = max(3, args.reps // (4 if b >= 64 else 1) // (3 if mode == "f32" and b >= 64 else 1))   # (f16 = the opt-in speed mode: the reference figure is fp16 too)

                def enc_only():
                    return eng.encode(feat, flen)

                def whole():
                    enc, elen = eng.encode(*eng.frontend(wav, wlen))
                    rows, flag = HipEngine.collect(eng.ctc_greedy(enc, elen))
                    assert not flag
                    return rows
                ms_enc, ms_all = timed(enc_only, reps), timed(whole, reps)
                enc, elen = enc_only()
                rec = {"batch": b, "seconds": sec, "mode": mode, "encoder_ms": round(ms_enc, 3), "whole_path_ms": round(ms_all, 3),
                       "encoder_frames": int(enc.shape[2]), "token_rows": int(b * enc.shape[2]), "reps": reps,
                       "rtfx_whole_path": round(b * sec / ms_all * 1e3, 1),
                       "reference_published_encoder_ms_cuda_fp16": PUBLISHED_MS[(b, sec)],
                       "hbm_in_use_gb": round((torch.cuda.mem_get_info(dev)[1] - torch.cuda.mem_get_info(dev)[0]) / 2**30, 2),
                       "note": "encoder = gam_encode on resident log-mel; whole path = log-mel + encoder + CTC greedy + ids on the host; "
                               "reference figure: /root/reference/evaluation.md:65-67 (unnamed CUDA GPU, fp16 autocast) -- context only"}
                f.write(json.dumps(rec) + "\n")
                f.flush()
                print(json.dumps({k: rec[k] for k in ("batch", "seconds", "mode", "encoder_ms", "whole_path_ms", "reference_published_encoder_ms_cuda_fp16")}), flush=True)
            eng.set_gemm_mode("f16x3")
            del wav, feat
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
