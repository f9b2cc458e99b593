#!/usr/bin/env python3
"""Experiment: one batch of B x 20 s as P independent part-batches on P HIP streams (P library handles), so that one
part's GEMM prologues / epilogues / LayerNorms / attention overlap the other parts' MFMA main loops.

    python tools/exp_two_streams.py [--batch 32] [--parts 1,2,4,1,2] [--reps 10]

Prints ms per B-utterance step for every P.  r02 (B = 32, power-capped): no gain.  r04 asks the same question at the
strong-scaling points (B = 4 / 8), where the step is latency-bound rather than power-bound."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gigaam_amd  # noqa: E402
from gigaam_amd import synth, workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--parts", default="1,2,4,1,2")
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ck = synth.make_checkpoint("v2_ctc", seed=0)
    B = args.batch
    wav, wlen = workloads.config2_batch(B, 20.0)
    wav, wlen = wav.to(dev), wlen.to(dev)
    for parts in [int(p) for p in args.parts.split(",")]:
        if B % parts:
            continue
        engs = [gigaam_amd.model_from_checkpoint(ck, dev).encoder.engine for _ in range(parts)]
        streams = [torch.cuda.Stream(dev) for _ in range(parts)]
        n = B // parts
        chunks = [(wav[i * n:(i + 1) * n].contiguous(), wlen[i * n:(i + 1) * n].contiguous()) for i in range(parts)]

        def step():
            outs = []
            for e, s, (w, l) in zip(engs, streams, chunks):
                with torch.cuda.stream(s):
                    enc, elen = e.encode(*e.frontend(w, l))
                    outs.append(e.ctc_greedy(enc, elen))
            return outs

        for _ in range(4):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.reps * 1e3
        print(f"{parts} stream(s) x {n} utterances: {ms:.3f} ms per {B} x 20 s ({B * 20.0 / ms * 1e3:.0f} x real time)", flush=True)
        del engs


if __name__ == "__main__":
    main()
