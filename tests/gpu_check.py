"""Stage-by-stage diagnostic of the HIP path against the CPU oracle (not a pytest file;
the pytest suite proper is tests/test_*.py).  Prints one line per stage with the max
abs error on valid frames and never stops at the first mismatch, so a single GPU-box
call localises a bug.    python tests/gpu_check.py [--out gpurun_out/check.json]
"""
import argparse
import json
import os
import sys
import time
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigaam_amd import synth  # noqa: E402
from gigaam_amd.engine import HipEngine, build_config  # noqa: E402
from oracle import gigaam_oracle as O  # noqa: E402

RES = {}


T0 = time.time()


def report(name, **kw):
    RES[name] = kw
    print(f"[{time.time() - T0:7.1f}s] [{name}] " + " ".join(f"{k}={v:.3e}" if isinstance(v, float) else f"{k}={v}" for k, v in kw.items()), flush=True)


def guarded(fn):
    def run(*a, **k):
        try:
            fn(*a, **k)
        except Exception as e:  # keep going
            traceback.print_exc()
            report(fn.__name__ + str(a[:1]), error=repr(e))
    return run


def valid_mask(t, lens):
    return torch.arange(t)[None, :] < lens[:, None].to(torch.int64)


@guarded
def check_gemm():
    ck = synth.make_checkpoint("v2_ctc", seed=1, n_layers=0)
    eng = HipEngine(build_config(ck["cfg"]["preprocessor"], ck["cfg"]["encoder"], None), {}, torch.device("cuda:0"))
    g = torch.Generator().manual_seed(0)
    for (m, n, k, act) in [(128, 128, 32, 0), (300, 200, 64, 0), (1000, 768, 768, 1), (257, 34, 768, 0), (515, 1536, 96, 2), (16064, 768, 3072, 0)]:
        a = torch.randn(m, k, generator=g)
        w = torch.randn(n, k, generator=g) / k ** 0.5
        b = torch.randn(n, generator=g)
        ref = a.double() @ w.double().t() + b.double()
        if act == 1:
            ref = ref * torch.sigmoid(ref)
        if act == 2:
            ref = torch.relu(ref)
        out = eng.op_gemm(a, w, b, act).cpu()
        report(f"gemm_{m}x{n}x{k}_act{act}", max_err=float((out.double() - ref).abs().max()), ref_absmax=float(ref.abs().max()))
    # timing of the flagship shapes
    for (m, n, k) in [(16064, 3072, 768), (16064, 768, 3072), (16064, 768, 768)]:
        a = torch.randn(m, k, device="cuda")
        w = torch.randn(n, k, device="cuda")
        eng.op_gemm(a, w)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            eng.op_gemm(a, w)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        report(f"gemm_time_{m}x{n}x{k}", ms=dt * 1e3, tflops=2.0 * m * n * k / dt / 1e12)


@guarded
def check_model(model, n_layers, batch, secs, lens, seed_audio=11, deep=True):
    ck = synth.make_checkpoint(model, seed=1, n_layers=n_layers)
    cfg, sd = ck["cfg"], ck["state_dict"]
    tag = f"{model}_l{n_layers}_b{batch}"
    wav, wlen = synth.synth_audio(batch, secs, seed=seed_audio, lengths=lens)
    eng = HipEngine(build_config(cfg["preprocessor"], cfg["encoder"], cfg.get("head")), sd, torch.device("cuda:0"))
    eng.set_gemm_mode(os.environ.get("GAM_CHECK_MODE", "f16x3"))
    with torch.no_grad():
        feat_o, flen_o = O.log_mel(wav, wlen, cfg["preprocessor"], sd["preprocessor.featurizer.0.spectrogram.window"],
                                   sd["preprocessor.featurizer.0.mel_scale.fb"])
        stages = {}
        enc_o, elen_o = O.encoder_forward(sd, cfg["encoder"], feat_o, flen_o, stages=stages)
    # frontend
    feat_g, flen_g = eng.frontend(wav, wlen)
    fm = valid_mask(feat_o.shape[2], flen_o)[:, None, :]
    report(tag + "/frontend", max_err=float(((feat_g.cpu() - feat_o) * fm).abs().max()), len_ok=flen_g.cpu().tolist() == flen_o.tolist(),
           shape_ok=tuple(feat_g.shape) == tuple(feat_o.shape))
    # encoder stages, fed with the ORACLE features so errors do not compound
    vm = valid_mask(enc_o.shape[2], elen_o)
    if deep:
        for nl in range(0, n_layers + 1):
            _, elen_g, tok = eng.encode(feat_o, flen_o, n_layers_run=nl, want_tokens=True)
            ref = stages["pre_encode"] if nl == 0 else stages[f"layer{nl - 1}"]
            err = ((tok.cpu() - ref) * vm[:, :, None]).abs()
            report(tag + f"/layers{nl}", max_err=float(err.max()), mean_err=float(err.mean()), finite=bool(torch.isfinite(tok).all()))
    enc_g, elen_g = eng.encode(feat_o, flen_o)
    err = ((enc_g.cpu() - enc_o) * vm[:, None, :]).abs()
    report(tag + "/encoded", max_err=float(err.max()), len_ok=elen_g.cpu().tolist() == elen_o.tolist(), dtype=str(elen_g.dtype))
    # end to end from the wav
    enc_e, elen_e = eng.encode(feat_g, flen_g)
    report(tag + "/encoded_e2e", max_err=float(((enc_e.cpu() - enc_o) * vm[:, None, :]).abs().max()))
    if cfg["head"]["_target_"].endswith("CTCHead"):
        with torch.no_grad():
            lp_o = O.ctc_log_probs(sd, enc_o)
            dec_o = O.ctc_greedy(lp_o, elen_o)
        lp_g = eng.ctc_head(enc_o).cpu()
        report(tag + "/ctc_log_probs", max_err=float(((lp_g - lp_o) * vm[:, :, None]).abs().max()))
        for src, enc_in, len_in in (("oracle_enc", enc_o, elen_o), ("hip_enc", enc_g, elen_g), ("hip_e2e", enc_e, elen_e)):
            ids, frames, counts = eng.ctc_greedy(enc_in, len_in)
            n = counts.cpu().tolist()
            got = [(ids[i, :c].cpu().tolist(), frames[i, :c].cpu().tolist()) for i, c in enumerate(n)]
            report(tag + f"/ctc_greedy[{src}]", exact=got == dec_o, counts=n, ref_counts=[len(a) for a, _ in dec_o])
    else:
        ms = cfg["decoding"]["max_symbols_per_step"]
        trace = []
        print(f"[{time.time() - T0:7.1f}s] oracle rnnt_greedy ...", flush=True)
        with torch.no_grad():
            dec_o = O.rnnt_greedy(sd, enc_o, elen_o, ms, trace=trace)
        print(f"[{time.time() - T0:7.1f}s] oracle rnnt_greedy done, {len(trace)} steps", flush=True)
        cap = max(sum(1 for t in trace if t[0] == b) for b in range(batch))
        for src, enc_in, len_in in (("oracle_enc", enc_o, elen_o), ("hip_enc", enc_g, elen_g)):
            ids, frames, counts, dump, dcount = eng.rnnt_greedy(enc_in, len_in, ms, dump_cap=cap)
            n = counts.cpu().tolist()
            got = [(ids[i, :c].cpu().tolist(), frames[i, :c].cpu().tolist()) for i, c in enumerate(n)]
            worst = 0.0
            if src == "oracle_enc" and got == dec_o:
                for b in range(batch):
                    ref = torch.stack([t[2] for t in trace if t[0] == b])
                    worst = max(worst, float((dump[b, : ref.shape[0]].cpu() - ref).abs().max()))
            report(tag + f"/rnnt_greedy[{src}]", exact=got == dec_o, counts=n, ref_counts=[len(a) for a, _ in dec_o],
                   logits_max_err=worst, steps=dcount.cpu().tolist())
    eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "check.json"))
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    print("device:", torch.cuda.get_device_name(0), flush=True)
    only = args.only
    if only in ("", "gemm"):
        check_gemm()
    if only in ("", "v2_ctc"):
        check_model("v2_ctc", 2, 3, 4.0, [64000, 50000, 33333])
        check_model("v2_ctc", 2, 1, 2.5, None, seed_audio=12)
    if only in ("", "v2_rnnt"):
        check_model("v2_rnnt", 2, 3, 4.0, [64000, 41234, 57000], seed_audio=13, deep=False)
    if only in ("", "v3_ctc"):
        check_model("v3_ctc", 2, 3, 4.0, [64000, 50000, 33333], seed_audio=14)
    if only in ("", "v3_e2e_rnnt"):
        check_model("v3_e2e_rnnt", 2, 2, 3.0, [48000, 30011], seed_audio=15, deep=False)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(RES, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
