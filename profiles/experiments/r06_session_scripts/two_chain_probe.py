"""EXPERIMENT (r06, VERDICT r5 #4 small grids): does a small batch finish sooner as K independent chains on K streams?

At 4 x 20 s the step is ~270 dependent launches of 10-40 us, a third to a half of each fixed cost (ramp, first k-tile latency, epilogue, drain):
the chip idles through every one of those.  K handles (replicated weights), each running batch / K utterances on its own stream, give the
hardware K independent launch chains to interleave.  This probe times  frontend + encoder + CTC greedy + ids on the host  for
batch in {2, 4, 8} utterances of 20 s as K = 1, 2, 4 chains, interleaved on one box, and checks the ids against the one-chain result.

    gpurun --timeout 900 -- 'python tools/two_chain_probe.py > gpurun_out/two_chain.txt'
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import gigaam_amd  # noqa: E402
from gigaam_amd import synth, workloads  # noqa: E402
from gigaam_amd.engine import HipEngine  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    kmax = 4
    ckpt = synth.make_checkpoint("v2_ctc", seed=0)
    models = [gigaam_amd.model_from_checkpoint(ckpt, dev) for _ in range(kmax)]
    engs = [m.encoder.engine for m in models]
    streams = [torch.cuda.Stream(dev) for _ in range(kmax)]

    def launch(eng, wav, wlen):
        feat, flen = eng.frontend(wav, wlen)
        enc, elen = eng.encode(feat, flen)
        return eng.ctc_greedy(enc, elen)

    def step(parts, k):
        decs = []
        if k == 1:
            decs.append(launch(engs[0], *parts[0]))
        else:
            for c in range(k):
                with torch.cuda.stream(streams[c]):
                    decs.append(launch(engs[c], *parts[c]))
        out = []
        for d in decs:
            rows, flag = HipEngine.collect(d)
            assert not flag
            out += rows
        return out

    batches = [int(x) for x in os.environ.get("PROBE_BATCHES", "2,4,8").split(",")]
    reps, steps = int(os.environ.get("PROBE_REPS", "3")), int(os.environ.get("PROBE_STEPS", "30"))
    for batch in batches:
        wav_h, wlen_h = workloads.config2_batch(batch, 20.0, first=0)
        wav, wlen = wav_h.to(dev), wlen_h.to(dev)
        want = None
        res = {}
        ks = [k for k in (1, 2, 4) if k <= batch]
        parts = {}
        for k in ks:
            n = batch // k
            parts[k] = [(wav[c * n:(c + 1) * n].contiguous(), wlen[c * n:(c + 1) * n].contiguous()) for c in range(k)]
        torch.cuda.synchronize()
        for rep in range(reps):
            for k in ks:
                for _ in range(8):
                    got = step(parts[k], k)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    got = step(parts[k], k)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / steps * 1e3
                res.setdefault(k, []).append(ms)
                if want is None:
                    want = got
                same = sum(1 for a, b in zip(want, got) if a == b)
                print(f"batch {batch} x 20 s  chains {k}  rep {rep}: {ms:.3f} ms per step  ids equal to one chain: {same}/{batch}", flush=True)
        print(f"SUMMARY batch {batch}: " + "  ".join(f"K={k}: {min(v):.3f} ms" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
