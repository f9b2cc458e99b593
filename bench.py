#!/usr/bin/env python3
"""Throughput bench of the MI355X hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the whole hot path over one batch of synthetic audio that is
already resident in HBM:  log-mel frontend -> 16-layer Conformer encoder -> CTC head +
greedy collapse  (BASELINE.json configs[1]: v2_ctc, batch 32 x 20 s per GPU), followed
by the only exchange the path has: an all-gather of the decoded (counts, ids, frames)
over RCCL.  Utterances are independent, so ranks shard the batch and scaling is weak
(32 utterances per GPU).  Weights are random-init tensors of the exact v2_ctc
architecture (gigaam_amd/synth.py; no checkpoints offline), arithmetic is fp32.

Rank 0 prints ONE JSON line.  Besides the driver's contract it carries
  roofline     -- the dominant kernel (the fp32 MFMA GEMM family, incl. the implicit-GEMM
                  stem conv), timed with HIP events on the launch stream inside the
                  timed region; achieved = algorithmic FLOP / event time.
  cpu_baseline -- the CPU oracle (a port of the reference's fp32 CPU path) timed on this
                  box's host cores on a bounded sample of the same workload (N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
F16_MFMA_PEAK_TFLOPS = 2500.0   # same guide: dense fp16/bf16 MFMA (32x32x16)
FLOP_PER_UTT_20S_V2 = 325.9e9   # SURVEY.md §8d / BASELINE.md §3


# ----------------------------------------------------------------------------- dist helpers
def world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def shard_range(n_items: int, rank: int, n_ranks: int):
    """Contiguous block of ceil(n/ranks) items per rank."""
    per = (n_items + n_ranks - 1) // n_ranks
    return min(n_items, rank * per), min(n_items, (rank + 1) * per)


def gather_decoded(counts: torch.Tensor, ids: torch.Tensor, frames: torch.Tensor):
    """The path's one exchange step: all-gather of the fixed-size padded decode buffers
    (<= 4 KB per utterance, latency-bound).  Rank-major order."""
    if world() == 1:
        return counts, ids, frames
    n = world()
    outs = []
    for t in (counts, ids, frames):
        buf = [torch.empty_like(t) for _ in range(n)]
        dist.all_gather(buf, t.contiguous())
        outs.append(torch.cat(buf, dim=0))
    return tuple(outs)


def max_over_ranks(seconds: float) -> float:
    if world() == 1:
        return seconds
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier_sync():
    if world() > 1:
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if world() > 1:
        dist.barrier()


# ----------------------------------------------------------------------------- cpu baseline
def cpu_baseline(ckpt, wav, wlen, n_utts: int, gpu_decoded):
    """CPU oracle (fp32, all host cores) on the first n_utts utterances; also reports
    whether the GPU ids/frames of those utterances are identical."""
    from oracle import gigaam_oracle as O
    # many small fp32 ops: beyond ~32 threads the oracle gets SLOWER on a many-core host
    # (256 threads measured 1.2x real time vs 8 threads ~24x), so cap and report the count
    threads = min(32, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    w, l = wav[:n_utts].cpu(), wlen[:n_utts].cpu()
    with torch.no_grad():
        O.transcribe_ids(ckpt, w[:1, :16000].contiguous(), torch.tensor([16000]))  # warm-up
        t0 = time.perf_counter()
        dec, _, _ = O.transcribe_ids(ckpt, w, l)
        dt = time.perf_counter() - t0
    audio_s = float(l.sum()) / 16000.0
    same = [list(a) == list(b) and list(c) == list(d) for (a, c), (b, d) in zip(dec, gpu_decoded[:n_utts])]
    return {
        "value": round(audio_s / dt, 3), "unit": "audio-sec/wall-sec", "cores": threads,
        "host_cpus": os.cpu_count(), "kind": "port",
        "sample": f"{n_utts} of the batch's utterances ({audio_s:.0f} s audio), oracle/gigaam_oracle.py fp32, {dt:.1f} s wall",
        "gpu_ids_identical": f"{sum(same)}/{len(same)}",
    }


# ----------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="v2_ctc")
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--layers", type=int, default=-1, help="debug only: fewer layers INVALIDATES the number")
    ap.add_argument("--cpu-utts", type=int, default=8, help="utterances in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--rnnt-blank-bias", type=float, default=None,
                    help="RNN-T models: blank bias of the synthetic joint (default: emission-heavy synthetic head)")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event timing")
    ap.add_argument("--gemm", default="f16x3", choices=["f16x3", "f32"],
                    help="dense-contraction arithmetic: split-fp16 MFMA (fp32-equivalent, default) or exact fp32 MFMA")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_ranks = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if n_ranks > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=n_ranks, device_id=torch.device("cuda", local_rank))
    assert n_ranks == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={n_ranks}"
    dev = torch.device("cuda", local_rank)

    import gigaam_amd
    from gigaam_amd import synth

    over = {} if args.layers < 0 else {"n_layers": args.layers}
    ckpt = synth.make_checkpoint(args.model, seed=0, rnnt_blank_bias=args.rnnt_blank_bias, **over)
    model = gigaam_amd.model_from_checkpoint(ckpt, dev)
    eng = model.encoder.engine
    eng.set_gemm_mode(args.gemm)
    is_ctc = ckpt["cfg"]["head"]["_target_"].endswith("CTCHead")
    max_sym = ckpt["cfg"]["decoding"].get("max_symbols_per_step", 10)

    # this rank's shard of the global batch: `batch` utterances, seeded by global index
    g0, g1 = shard_range(args.batch * n_ranks, rank, n_ranks)
    wav_h, wlen_h = synth.synth_audio(g1 - g0, args.seconds, seed=1000 + rank)
    wav, wlen = wav_h.to(dev), wlen_h.to(dev)          # resident in HBM before the timed region

    def step():
        feat, flen = eng.frontend(wav, wlen)
        enc, elen = eng.encode(feat, flen)
        if is_ctc:
            ids, frames, counts = eng.ctc_greedy(enc, elen)
        else:
            ids, frames, counts = eng.rnnt_greedy(enc, elen, max_sym)
        return gather_decoded(counts, ids, frames)

    for _ in range(args.warmup):
        step()
    barrier_sync()
    if not args.no_profile:
        eng.profile_enable(2)      # inside the timed region: events around the GEMM family only
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier_sync()
    dt = max_over_ranks(time.perf_counter() - t0)
    prof = prof_all = None
    if not args.no_profile:
        prof = eng.profile_read()
        eng.profile_enable(1)      # one extra, untimed step for the per-class breakdown
        step()
        torch.cuda.synchronize()
        prof_all = eng.profile_read()
        eng.profile_enable(0)

    if rank != 0:
        if n_ranks > 1:
            dist.destroy_process_group()
        return

    audio_s = args.seconds * args.batch * n_ranks * args.steps
    ms_step = dt / args.steps * 1e3
    counts, ids, frames = out
    n_c = counts.cpu().tolist()
    decoded = [(ids[i, :c].cpu().tolist(), frames[i, :c].cpu().tolist()) for i, c in enumerate(n_c[: g1 - g0])]
    line = {
        "metric": f"RTFx {args.model} batch{args.batch}x{args.seconds:g}s (log-mel + encoder + greedy decode)",
        "value": round(audio_s / dt, 1), "unit": "audio-sec/wall-sec", "n_gpus": n_ranks, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32" if args.gemm == "f32" else "f32 (GEMMs: 3-term split on fp16 MFMA, fp32 accumulate)",
        "data": "synthetic",
        "config": {"workload": f"{args.model} (16-layer Conformer, random-init weights), {args.batch} x {args.seconds:g} s "
                               f"16 kHz utterances per GPU, frontend+encoder+{'CTC' if is_ctc else 'RNN-T'} greedy, "
                               "final all-gather of ids", "global_batch": args.batch * n_ranks,
                   "audio_seconds_per_step": args.seconds * args.batch * n_ranks,
                   "parallelism": f"dp{n_ranks} (utterance shards, RCCL all-gather of ids only)"},
        "encoder_ms_per_utt": round(ms_step / args.batch, 4),
        "tokens_decoded_per_step": int(sum(n_c)),
    }
    if args.layers >= 0:
        line["INVALID"] = "debug run with --layers"
    if prof is not None:
        fam = ("gemm", "conv2")  # one kernel template: gam_gemm_f32_kernel<ACT> (plain + implicit-GEMM stem conv)
        flop = sum(prof[k]["work"] for k in fam)
        ms = sum(prof[k]["ms"] for k in fam)
        n = sum(prof[k]["launches"] for k in fam)
        ach = flop / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        if args.gemm == "f32":
            kern, peak, peak_note = "gam_gemm_f32_kernel (v_mfma_f32_32x32x2_f32; plain + implicit-GEMM conv)", FP32_MFMA_PEAK_TFLOPS, \
                "fp32 dense MFMA peak"
        else:
            # every algorithmic FLOP costs three fp16 MFMA FLOPs (hi.hi + hi.lo + lo.hi): the ceiling of
            # this arithmetic is a third of the fp16 dense peak
            kern = ("gam_gemm_sp_kernel (LDS-DMA, sp32 operands; small GEMMs: gam_gemm_f16x3_kernel) -- 3x "
                    "v_mfma_f32_32x32x16_f16 per product; plain + implicit-GEMM conv")
            peak, peak_note = F16_MFMA_PEAK_TFLOPS / 3.0, "fp16 dense MFMA peak (2500) / 3 issued MFMA FLOP per algorithmic FLOP"
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", f"pmc_traffic_{args.gemm}.json")
        if args.model == "v2_ctc" and args.batch == 32 and args.seconds == 20.0 and os.path.exists(tpath):
            tj = json.load(open(tpath))       # committed rocprofv3 PMC passes of this same command
            traffic, traffic_src = round(tj["traffic_bytes_per_launch"]), os.path.relpath(tpath, ROOT)
        alg_bytes = sum(prof[k].get("bytes", 0.0) for k in fam)
        line["roofline"] = {
            "kernel": kern, "bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
            "peak_note": peak_note, "issued_mfma_tflops": round(ach * (1.0 if args.gemm == "f32" else 3.0), 1),
            "algorithmic_bytes_per_launch": round(alg_bytes / max(1, n)),
            "launches_per_step": n // max(1, args.steps), "avg_launch_ms": round(ms / max(1, n), 4),
            "algorithmic_gflop_per_step": round(flop / args.steps / 1e9, 1),
            "share_of_step_time": round(ms / args.steps / ms_step, 3),
        }
        line["kernel_classes_ms_per_step"] = {k: round(v["ms"], 3) for k, v in prof_all.items() if v["launches"]}
        line["kernel_classes_note"] = "HIP-event time per class from one extra untimed step"
        whole = FLOP_PER_UTT_20S_V2 * (args.seconds / 20.0) * args.batch * n_ranks / (ms_step * 1e-3) / 1e12
        line["whole_path_tflops"] = round(whole, 2)
    if n_ranks == 1 and args.cpu_utts > 0:
        try:
            line["cpu_baseline"] = cpu_baseline(ckpt, wav_h, wlen_h, min(args.cpu_utts, g1 - g0), decoded)
        except Exception as e:  # the bench line must still print
            line["cpu_baseline"] = {"error": repr(e)}
    print(json.dumps(line, ensure_ascii=False), flush=True)
    if n_ranks > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
