#!/usr/bin/env python3
"""Experiment: the 32 x 20 s batch as TWO independent half-batches on two HIP streams (two library handles), so that
one half's GEMM prologues / epilogues / LayerNorms / attention overlap the other half's MFMA main loops.
Prints ms per 32-utterance step for 1 stream x 32 and 2 streams x 16 (and 4 x 8)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gigaam_amd  # noqa: E402
from gigaam_amd import synth, workloads  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    ck = synth.make_checkpoint("v2_ctc", seed=0)
    wav, wlen = workloads.config2_batch(32, 20.0)
    wav, wlen = wav.to(dev), wlen.to(dev)
    for parts in (1, 2, 4, 1, 2):
        engs = [gigaam_amd.model_from_checkpoint(ck, dev).encoder.engine for _ in range(parts)]
        streams = [torch.cuda.Stream(dev) for _ in range(parts)]
        n = 32 // parts
        chunks = [(wav[i * n:(i + 1) * n].contiguous(), wlen[i * n:(i + 1) * n].contiguous()) for i in range(parts)]

        def step():
            outs = []
            for e, s, (w, l) in zip(engs, streams, chunks):
                with torch.cuda.stream(s):
                    enc, elen = e.encode(*e.frontend(w, l))
                    outs.append(e.ctc_greedy(enc, elen))
            return outs

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = 10
        for _ in range(k):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / k * 1e3
        print(f"{parts} stream(s) x {n} utterances: {ms:.2f} ms per 32 x 20 s ({640.0 / ms * 1e3:.0f} x real time)", flush=True)
        del engs


if __name__ == "__main__":
    main()
