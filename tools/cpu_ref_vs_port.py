#!/usr/bin/env python3
"""The reference's OWN modules timed beside the oracle port on the same utterances and thread count -- runs only where
/root/reference exists (the build container).  bench.py's cpu_baseline leg times the port (kind "port") because the
reference tree does not travel to the GPU box; this script states how far the port is from the real thing.

    python tools/cpu_ref_vs_port.py [--utts 4] [--threads 8]   ->  profiles/r03_cpu_ref_vs_port.json

Both legs: log-mel by the oracle's restatement (the reference's torchaudio frontend is not installed), then
ConformerEncoder.forward + CTCHead + CTCGreedyDecoding.decode (reference gigaam/encoder.py:605-647, decoder.py:18-21,
decoding.py:56-96) vs oracle.encoder_forward + ctc_log_probs + ctc_greedy, calls of 2 utterances like bench.py."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from gigaam_amd import synth, workloads  # noqa: E402
from oracle import gigaam_oracle as O  # noqa: E402
from oracle.ref_shim import import_reference  # noqa: E402
from make_golden import kw, strip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=4)
    ap.add_argument("--threads", type=int, default=min(32, os.cpu_count() or 1))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_cpu_ref_vs_port.json"))
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    ref = import_reference()
    ck = synth.make_checkpoint("v2_ctc", seed=0)
    cfg, sd = ck["cfg"], ck["state_dict"]
    wav, wlen = workloads.config2_batch(args.utts, 20.0, rank=0)
    enc = ref.encoder.ConformerEncoder(**kw(cfg["encoder"])).eval()
    enc.load_state_dict(strip(sd, "encoder."))
    head = ref.decoder.CTCHead(**kw(cfg["head"])).eval()
    head.load_state_dict(strip(sd, "head."))
    dec = ref.decoding.CTCGreedyDecoding(cfg["decoding"]["vocabulary"])

    def feats(w, l):
        return O.log_mel(w, l, cfg["preprocessor"], sd["preprocessor.featurizer.0.spectrogram.window"], sd["preprocessor.featurizer.0.mel_scale.fb"])

    def run_ref(w, l):
        f, fl = feats(w, l)
        y, yl = enc(f, fl)
        return [(i, fr) for _t, i, fr in dec.decode(head, y, yl)]

    def run_port(w, l):
        return O.transcribe_ids(ck, w, l)[0]

    out = {}
    with torch.no_grad():
        for name, fn in (("reference", run_ref), ("port", run_port)):
            fn(wav[:1, :16000].contiguous(), torch.tensor([16000]))     # warm-up
            best, ids = None, None
            for _ in range(2):
                t0 = time.perf_counter()
                res = []
                for i in range(0, args.utts, 2):
                    res += fn(wav[i:i + 2], wlen[i:i + 2])
                dt = time.perf_counter() - t0
                if best is None or dt < best:
                    best, ids = dt, res
            out[name] = (best, ids)
            print(name, f"{best:.2f} s", flush=True)
    audio_s = float(wlen.sum()) / 16000.0
    same = sum(list(a[0]) == list(b[0]) and list(a[1]) == list(b[1]) for a, b in zip(out["reference"][1], out["port"][1]))
    rec = {
        "what": "reference gigaam ConformerEncoder + CTCHead + CTCGreedyDecoding (unmodified modules, oracle/ref_shim.py) vs "
                "oracle/gigaam_oracle.py on the same utterances of BASELINE config 2; log-mel by the oracle's frontend in both legs",
        "utterances": args.utts, "audio_seconds": audio_s, "threads": args.threads, "host_cpus": os.cpu_count(),
        "reference_seconds": round(out["reference"][0], 3), "port_seconds": round(out["port"][0], 3),
        "reference_rtfx": round(audio_s / out["reference"][0], 2), "port_rtfx": round(audio_s / out["port"][0], 2),
        "port_over_reference": round(out["reference"][0] / out["port"][0], 3),
        "ids_identical": f"{same}/{args.utts}", "where": "build container (no GPU); best of 2 passes per leg",
    }
    json.dump(rec, open(args.out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
