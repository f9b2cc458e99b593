#!/usr/bin/env python3
"""Throughput bench of the MI355X hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W                      # the driver's command: config 2, weak scaling
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
    python bench.py --config {1,2,3,4,5} [--scaling weak|strong] [--gather rccl|torch] [--gemm f16x3|f32]

One step = one pass of the whole hot path over one batch (config 4/5: one pass over the rank's share of the
set) of synthetic audio: log-mel frontend -> 16-layer Conformer encoder -> greedy decode -> decoded ids on the
HOST (the D2H copy the product's transcribe path always pays), followed by the only exchange the path has: a
gather of the decoded (index, counts, ids, frames) over RCCL (gam_gather_ids, include/gigaam_hip.h).

  config 2 (default, the headline; BASELINE.json configs[1])  v2_ctc, 32 x 20 s
  config 3  v2_rnnt, same audio, blank-dominant synthetic head (tests/golden/fullsize_meta.json)
  config 1  v2_ctc, one 5 s clip
  config 4  v3_e2e_rnnt (V = 1025), 128 utterances per GPU (weak; 1024 at 8 GPUs) or 1024 in total (strong),
            durations U(5 s, 20 s), sorted into 32-utterance batches dealt to the ranks
  config 5  v2_ctc longform: 1 h of audio, the reference's chunk packer, the chunks dealt by duration (LPT) to the ranks,
            each rank's share in length-sorted batches of 16 streamed through the pinned double-buffered feeder (always
            strong scaling; the file order is restored in the result)
--scaling weak: per-GPU work fixed (32 utterances per GPU); strong: the global batch of 32 split over the ranks.

Utterances are independent, so ranks own disjoint utterances with replicated weights.  Weights are random-init
tensors of the exact architecture (gigaam_amd/synth.py; no checkpoints offline), arithmetic is fp32.

Rank 0 prints ONE JSON line.  Besides the driver's contract it carries
  roofline            the dominant kernel family (the GEMMs, incl. the implicit-GEMM stem conv), timed with HIP
                      events on the launch stream inside the timed region; achieved = algorithmic FLOP / event time
  roofline_f32_exact  the same steps re-timed with the dense contractions on exact-fp32 MFMA (the reference's own
                      arithmetic), so that figure is on the driver's clock too
  cpu_baseline        the REFERENCE's own modules (oracle/_ref, built by oracle/build_ref.py from /root/reference) on this
                      box's host cores (N=1 only), ids compared with the GPU's for every utterance of the sample
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

from gigaam_amd import shard  # noqa: E402
from gigaam_amd.shard import shard_range  # noqa: E402,F401  (re-exported: tests/test_distributed_gloo.py)

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
F16_MFMA_PEAK_TFLOPS = 2500.0   # same guide: dense fp16/bf16 MFMA (32x32x16)
PEAK_SCLK_MHZ = 2400.0          # same guide: the boost clock the peaks are quoted at
FLOP_PER_UTT_20S_V2 = 325.9e9   # SURVEY.md §8d / BASELINE.md §3


# ----------------------------------------------------------------------------- dist helpers
def world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def gather_decoded(counts: torch.Tensor, ids: torch.Tensor, frames: torch.Tensor):
    """torch.distributed form of the path's one exchange (rank-major); see shard.torch_gather."""
    _, c, i, f = shard.torch_gather(None, counts, ids, frames)
    return c, i, f


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


def make_gather(kind: str, rank: int, n_ranks: int, dev: torch.device, strict: bool = False):
    """The exchange callable (index, counts, ids, frames) -> the same, rank-major.  "rccl": gam_gather_ids behind
    the C ABI (the RCCL id travels over torch.distributed's store); "torch": dist.all_gather (cross-check)."""
    if n_ranks == 1:
        return (lambda index, counts, ids, frames: (index, counts, ids, frames)), "none (one rank)"
    if kind == "rccl":
        # the 128-byte RCCL id travels over torch.distributed's store, on the MAIN thread (a collective of the default group)
        import ctypes as C
        import threading
        from gigaam_amd import _lib
        buf = C.create_string_buffer(128)
        rc0 = _lib.load_library().gam_comm_unique_id(buf) if rank == 0 else 0
        sent = [buf.raw if (rank == 0 and rc0 == 0) else b""]
        dist.broadcast_object_list(sent, src=0)
        uid = sent[0]
        # gam_comm_create (ncclCommInitRank behind the C ABI) has only ever run with a world of one on the builder's 1-GPU
        # boxes: it gets a watchdog thread, so that a rendezvous that never completes costs this run its RCCL-behind-the-ABI
        # gather, not its bench line (the ranks then agree, below, to use torch.distributed's own RCCL group).
        box, ok = {}, 0

        def create():
            try:
                torch.cuda.set_device(dev)       # (the current device is per thread)
                box["comm"] = shard.HipComm(rank, n_ranks, dev, lambda _mine: uid)
            except Exception as e:   # noqa: BLE001
                box["err"] = e
        th = threading.Thread(target=create, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("GAM_COMM_TIMEOUT_S", "90")))
        if th.is_alive():
            print(f"[bench] rank {rank}: gam_comm_create did not return within the watchdog; using torch.distributed for the gather", file=sys.stderr)
        elif "err" in box:   # the line must still be produced: fall back to torch's own RCCL group, and say so
            print(f"[bench] rank {rank}: gam_comm_create failed ({box['err']}); using torch.distributed for the gather", file=sys.stderr)
        else:
            comm, ok = box["comm"], 1
        flag = torch.tensor([ok], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)      # every rank takes the same path
        if int(flag.item()) == 1:
            return comm.gather, "gam_gather_ids (RCCL ncclAllGather behind the C ABI)"
        if strict:   # tools/scale8.sh: the day-one run must say loudly that the exchange behind the C ABI did not come up
            raise SystemExit(f"[bench] rank {rank}: --strict-gather and gam_comm_create did not succeed on every rank "
                             f"({box.get('err', 'watchdog expired' if th.is_alive() else 'another rank failed')})")

    on = dev if dist.get_backend() == "nccl" else torch.device("cpu")

    def tg(index, counts, ids, frames):
        mv = lambda t: None if t is None else t.to(on)  # noqa: E731
        return shard.torch_gather(mv(index), mv(counts), mv(ids), mv(frames))
    return tg, f"torch.distributed all_gather ({'RCCL' if on.type == 'cuda' else 'gloo: shared-device rehearsal'})"



# ----------------------------------------------------------------------------- board power / clock (amdgpu hwmon)
def _hwmon_dir(dev_index: int):
    """The amdgpu hwmon directory of torch device ``dev_index`` (matched by PCI address), or None."""
    import glob
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
    except Exception:
        want = None
    cands = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        hw = sorted(glob.glob(os.path.join(d, "hwmon", "hwmon*")))
        if not hw:
            continue
        real = os.path.basename(os.path.realpath(d))
        cands.append((real, hw[0]))
    for real, hw in cands:
        if want and real.startswith(want):
            return hw
    return cands[0][1] if len(cands) == 1 else None


class PowerSampler:
    """Samples board power (power1_average / power1_input, microwatts) and the shader clock (freq1_input, Hz) from the
    amdgpu hwmon files while a few untimed steps run: the measurement behind DESIGN's "the GEMM loop is power-limited"."""

    def __init__(self, dev_index: int, period_s: float = 0.01):
        import threading
        import shutil
        self.hw = _hwmon_dir(dev_index)
        self.smi = None if self.hw else (shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None))
        self.raw = None
        self.period = period_s
        self.samples = []
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except Exception:
            return None

    def _run_smi(self):
        """Fallback when sysfs is hidden: poll ``rocm-smi -P -c --json`` (a few samples per second)."""
        import re
        import subprocess
        while not self._stop.is_set():
            try:
                txt = subprocess.run([self.smi, "-P", "-c", "--json"], capture_output=True, text=True, timeout=5).stdout
                card = next(iter(json.loads(txt).values()))
                if self.raw is None:
                    self.raw = card
                w = next((float(v) for k, v in card.items() if "ower" in k and "(W)" in k and re.fullmatch(r"[0-9.]+", str(v))), None)
                f = next((float(re.search(r"([0-9.]+)\s*Mhz", str(v), re.I).group(1)) for k, v in card.items()
                          if k.lower().startswith("sclk") and re.search(r"[0-9.]+\s*Mhz", str(v), re.I)), None)
                self.samples.append((w * 1e6 if w else None, f * 1e6 if f else None))
            except Exception:
                self.samples.append((None, None))

    def _run(self):
        if not self.hw:
            return self._run_smi()
        pw = next((p for p in ("power1_average", "power1_input") if os.path.exists(os.path.join(self.hw, p))), None)
        while not self._stop.is_set():
            w = self._read(os.path.join(self.hw, pw)) if pw else None
            f = self._read(os.path.join(self.hw, "freq1_input"))
            self.samples.append((w, f))
            time.sleep(self.period)

    def __enter__(self):
        if self.hw or self.smi:
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self.hw or self.smi:
            self._th.join(timeout=6.0)

    def summary(self):
        if not self.hw and not self.smi:
            return {"available": False, "note": "neither an amdgpu hwmon directory nor rocm-smi is visible here"}
        ws = [w * 1e-6 for w, _ in self.samples if w]
        fs = [f * 1e-6 for _, f in self.samples if f]
        cap = self._read(os.path.join(self.hw, "power1_cap")) if self.hw else None
        out = {"available": bool(ws or fs), "samples": len(self.samples), "source": self.hw or (self.smi + " -P -c --json")}
        if not ws and not fs and self.raw is not None:
            out["unparsed_sample"] = {k: str(v)[:40] for k, v in list(self.raw.items())[:12]}
        if ws:
            out.update(avg_w=round(sum(ws) / len(ws), 1), max_w=round(max(ws), 1))
        if cap:
            out["cap_w"] = round(cap * 1e-6, 1)
        if fs:
            out.update(avg_sclk_mhz=round(sum(fs) / len(fs)), min_sclk_mhz=round(min(fs)), max_sclk_mhz=round(max(fs)))
        return out


# ----------------------------------------------------------------------------- cpu baseline
def _ref_margin_of(model, is_rnnt, enc, elen, i):
    """Smallest top-1 / top-2 log-prob margin the REFERENCE saw while decoding utterance i (untimed; only called for an
    utterance whose GPU ids differ, to tell a near-tie of a random-weight head from a real disagreement)."""
    n = int(elen[i])
    e1 = enc[i:i + 1, :, :n].contiguous()
    if not is_rnnt:
        lp = model.head(e1)
        t2 = lp.topk(2, dim=-1).values
        return float((t2[..., 0] - t2[..., 1]).min())
    rec, orig = [], model.head.joint.joint

    def spy(f, g):
        out = orig(f, g)
        t2 = out.reshape(-1, out.shape[-1]).topk(2, dim=-1).values
        rec.append(float((t2[:, 0] - t2[:, 1]).min()))
        return out
    model.head.joint.joint = spy
    try:
        model.decoding.decode(model.head, e1, elen[i:i + 1])
    finally:
        del model.head.joint.joint
    return min(rec) if rec else None


def _ref_logprobs_of(model, is_rnnt, enc, elen, i):
    """The REFERENCE's log-probs for utterance i: CTC [n, V] per frame; RNN-T [steps, V], one row per joint evaluation in the order its
    greedy loop makes them (gigaam/decoding.py:162-205 through a recorder on RNNTJoint.joint)."""
    n = int(elen[i])
    e1 = enc[i:i + 1, :, :n].contiguous()
    if not is_rnnt:
        return model.head(e1)[0, :n].clone()
    rec, orig = [], model.head.joint.joint

    def spy(f, g):
        out = orig(f, g)
        rec.append(out.reshape(-1, out.shape[-1])[0].clone())
        return out
    model.head.joint.joint = spy
    try:
        model.decoding.decode(model.head, e1, elen[i:i + 1])
    finally:
        del model.head.joint.joint
    return torch.stack(rec) if rec else torch.zeros((0, 1))


def _mismatch_report(ref_lp, gpu_lp):
    """Walk the two log-prob sequences (frames for CTC, joint evaluations for RNN-T: identical steps until the first differing
    decision) up to AND INCLUDING the first step whose argmax differs: the largest |GPU - reference| over those rows, and the
    reference's own top-1 / top-2 margin on the diverging row -- a near-tie inside the 1e-3 logit bar, or a real disagreement."""
    n = min(int(ref_lp.shape[0]), int(gpu_lp.shape[0]))
    if n == 0:
        return {"steps_compared": 0}
    a, b = ref_lp[:n].float(), gpu_lp[:n].float()
    diff = (a - b).abs().amax(dim=1)
    am = a.argmax(dim=1) != b.argmax(dim=1)
    first = int(am.nonzero()[0]) if bool(am.any()) else n - 1
    t2 = a[first].topk(2).values
    return {"first_differing_step": first if bool(am.any()) else None, "steps_compared": first + 1,
            "max_logit_diff": float(diff[:first + 1].max()), "reference_margin_at_that_step": float(t2[0] - t2[1])}


def cpu_baseline(ckpt, batches, gpu_decoded, sweep: bool, equal_lengths: bool, port_too: bool, gpu_probe=None, ref_ids_out=None):
    """The reference's OWN modules (oracle/_ref: CPython bytecode of /root/reference/gigaam/*.py made by oracle/build_ref.py,
    imported through oracle/ref_shim.py) on this box's host cores, fp32: ``GigaAM.forward`` (FeatureExtractor ->
    ConformerEncoder, gigaam/model.py:27-37) + ``decoding.decode(head, ...)`` (model.py:96-124) = the body of
    ``transcribe()`` (model.py:126-140) behind ``load_audio``, batched.  ``batches`` = [(wav [b,L], wlen [b])] host tensors,
    each exactly as the GPU decoded it (a ragged batch must be seen whole: the reference's features of an utterance's last
    frames depend on its row's padding); equal-length batches are fed two utterances per call (larger CPU batches run slower
    per utterance).  With ``sweep`` the thread count is the best of 8 / 16 / 32 / 64 on a two-utterance sample.  Reports how
    many utterances decode to exactly the GPU's ids AND frames; for a mismatch, the reference's own smallest top-1/top-2
    margin on that utterance.  ``port_too``: oracle/gigaam_oracle.py (the restatement the tests use) timed beside it.
    Falls back to the port alone (kind = "port") only when oracle/_ref is missing, and says so."""
    from oracle import gigaam_oracle as O
    from oracle import ref_shim
    ncpu = os.cpu_count() or 1
    is_rnnt = "RNNT" in ckpt["cfg"]["decoding"]["_target_"]
    use_ref = ref_shim.reference_available()
    model = ref_shim.reference_model(ckpt) if use_ref else None

    def run_ref(w, l):
        with torch.inference_mode():
            enc, elen = model.forward(w, l)
            dec = model.decoding.decode(model.head, enc, elen)
        return [(list(i), list(f)) for _, i, f in dec], enc, elen

    def run_port(w, l):
        with torch.no_grad():
            dec, enc, elen = O.transcribe_ids(ckpt, w, l)
        return [(list(i), list(f)) for i, f in dec], enc, elen

    run = run_ref if use_ref else run_port
    calls = []                      # (wav, wlen) per CPU call
    for w, l in batches:
        w, l = w.cpu(), l.cpu()
        if equal_lengths:
            calls += [(w[i:i + 2].contiguous(), l[i:i + 2].contiguous()) for i in range(0, w.shape[0], 2)]
        else:
            calls.append((w, l))
    n_utts = sum(int(l.shape[0]) for _, l in calls)
    audio_s = float(sum(int(l.sum()) for _, l in calls)) / 16000.0
    sweep_out = {}
    best = min(32, ncpu)
    torch.set_num_threads(best)
    run(calls[0][0][:1, :16000].contiguous(), torch.tensor([16000]))  # warm-up
    if sweep:
        ws, ls = calls[0][0][:2], calls[0][1][:2]
        for th in sorted({min(8, ncpu), min(16, ncpu), min(32, ncpu), min(64, ncpu)}):
            torch.set_num_threads(th)
            t0 = time.perf_counter()
            run(ws, ls)
            sweep_out[th] = round(float(ls.sum()) / 16000.0 / (time.perf_counter() - t0), 2)
        best = max(sweep_out, key=sweep_out.get)
    torch.set_num_threads(best)
    t0 = time.perf_counter()
    outs = [run(w, l) for w, l in calls]
    dt = time.perf_counter() - t0
    dec = [d for o in outs for d in o[0]]
    same = [a == (list(b[0]), list(b[1])) for a, b in zip(dec, gpu_decoded[:n_utts])]
    if ref_ids_out is not None:
        ref_ids_out.update({i: d for i, d in enumerate(dec)})
    out = {
        "value": round(audio_s / dt, 3), "unit": "audio-sec/wall-sec", "cores": best, "host_cpus": ncpu,
        "kind": "reference" if use_ref else "port",
        "sample": (f"{n_utts} utterances ({audio_s:.0f} s audio) of the timed workload in {len(calls)} call(s) "
                   f"({'two equal-length utterances per call' if equal_lengths else 'each ragged batch whole, as the GPU decoded it'}), "
                   f"{dt:.1f} s wall, fp32, torch {torch.__version__} CPU"),
        "gpu_ids_identical": f"{sum(same)}/{len(same)}",
    }
    if use_ref:
        out["what"] = ("the reference's own GigaAMASR.forward + decoding.decode (= transcribe() behind load_audio; "
                       f"oracle/_ref {ref_shim.import_reference().kind} of /root/reference/gigaam, sha256 in oracle/ref_manifest.json); "
                       "torchaudio.transforms.MelSpectrogram is a stand-in on torch.stft (not installed here: row a1)")
    else:
        out["what"] = "oracle/gigaam_oracle.py (fp32 port of the reference's CPU path): oracle/_ref is MISSING on this box -- run oracle/build_ref.py"
    if sweep_out:
        out["thread_sweep_rtfx"] = {str(k): v for k, v in sweep_out.items()}
    if sum(same) != len(same):
        bad = [i for i, s in enumerate(same) if not s]
        out["mismatching_utterances"] = bad
        if use_ref:     # near-tie or real?  (row i of the concatenated calls -> its call and row)
            rows, k = {}, 0
            for ci, (_, l) in enumerate(calls):
                for r in range(int(l.shape[0])):
                    rows[k] = (ci, r)
                    k += 1
            with torch.inference_mode():
                out["mismatch_reference_min_margin"] = {str(i): _ref_margin_of(model, is_rnnt, outs[rows[i][0]][1], outs[rows[i][0]][2], rows[i][1])
                                                        for i in bad[:8]}
                if gpu_probe is not None:
                    # for every utterance whose ids differ, the logit difference GPU vs reference up to the diverging
                    # decision (north_star: RNN-T logits within 1e-3), and the same utterance decoded at the serial path's cluster size
                    rep, worst = {}, 0.0
                    for i in bad[:8]:
                        ref_lp = _ref_logprobs_of(model, is_rnnt, outs[rows[i][0]][1], outs[rows[i][0]][2], rows[i][1])
                        gpu_lp, extra = gpu_probe(i)
                        r = _mismatch_report(ref_lp, gpu_lp)
                        r.update(extra)
                        rep[str(i)] = r
                        worst = max(worst, r.get("max_logit_diff", 0.0))
                    out["mismatch_logits"] = rep
                    out["mismatch_max_logit_diff"] = worst
    if use_ref and port_too:
        t0 = time.perf_counter()
        pdec = [d for w, l in calls for d in run_port(w, l)[0]]
        dtp = time.perf_counter() - t0
        out["port"] = {"value": round(audio_s / dtp, 3), "cores": best, "wall_s": round(dtp, 1),
                       "ids_identical_to_reference": f"{sum(a == b for a, b in zip(pdec, dec))}/{len(dec)}",
                       "what": "oracle/gigaam_oracle.py on the same calls (the restatement the parity tests use)"}
    return out


# ----------------------------------------------------------------------------- workloads
def rnnt_bias_for(model_name: str, override):
    """Blank bias of the synthetic RNN-T head: the blank-dominant value the full-size goldens were made with."""
    if override is not None:
        return override
    meta = os.path.join(ROOT, "tests", "golden", "fullsize_meta.json")
    if os.path.exists(meta):
        m = json.load(open(meta)).get(f"fullsize_{model_name}")
        if m:
            return m.get("blank_bias")
    return None


def ragged_host(dec):
    """What a decode call returned (engine.Decoded) -> host lists: the blocking D2H + slicing of gigaam_amd.decoding._ragged
    (the split-fp16 range flag rides on the counts copy: engine.collect)."""
    from gigaam_amd.engine import HipEngine
    rows, flag = HipEngine.collect(dec)
    if flag:   # the product would repeat the batch in fp32 (model._with_f32_fallback); a timed step must not do that silently
        raise RuntimeError("split-fp16 range flag set during a bench step")
    return rows


def free_port() -> int:
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(n: int, argv) -> int:
    """``python bench.py --gpus N`` with no launcher around it (WORLD_SIZE unset): start the N ranks ourselves -- one
    process per GPU, the same environment contract torch.distributed.run provides (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT on 127.0.0.1) -- and pass rank 0's stdout through, so the caller still reads ONE JSON line.
    Returns the exit code (first non-zero rank's)."""
    import subprocess
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GAM_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                c = procs[r].poll()
                if c is None:
                    continue
                pending.discard(r)
                if c != 0 and rc == 0:
                    rc = c
                    print(f"[bench] rank {r} exited with code {c}; stopping the other ranks", file=sys.stderr)
                    for q in pending:
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return rc


def launch_selftest(rank: int, n_ranks: int, args):
    """--launch-selftest: the rank plumbing of this script without a GPU (gloo): rendezvous, the barrier-bracketed timed
    region with max-over-ranks, the exchange, ONE line from rank 0.  tests/test_distributed_gloo.py runs it through the
    same ``python bench.py --gpus 2`` entry the driver uses."""
    if n_ranks > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=n_ranks)
    counts = torch.tensor([rank + 1], dtype=torch.int32)
    ids = torch.full((1, 4), rank, dtype=torch.int32)
    barrier_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (rank + 1))
        _, gc, gids, _ = shard.torch_gather(None, counts, ids, ids)
    barrier_sync()
    dt = max_over_ranks(time.perf_counter() - t0)
    if rank == 0:
        emit({"selftest": True, "n_gpus": n_ranks, "steps": args.steps, "ms_per_step": round(dt / args.steps * 1e3, 3),
              "gathered_counts": gc.tolist(), "gathered_ids": gids[:, 0].tolist(),
              "launcher": "bench.py self-launch" if os.environ.get("GAM_BENCH_SELF_LAUNCHED") else "external"}, n_ranks)
    elif n_ranks > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=[1, 2, 3, 4, 5], help="BASELINE.json configuration (default 2: the headline)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--gather", default="rccl", choices=["rccl", "torch"], help="the final exchange: gam_gather_ids (C ABI) or torch.distributed")
    ap.add_argument("--comm-selftest", action="store_true", help="create the communicator behind the C ABI at world --gpus and run ONE gam_gather_ids "
                    "of a tiny buffer, then exit (tools/scale8.sh runs it first)")
    ap.add_argument("--strict-gather", action="store_true", help="N > 1: fail instead of falling back to torch.distributed when "
                    "gam_comm_create (RCCL behind the C ABI) does not come up on every rank")
    ap.add_argument("--model", default=None, help="override the configuration's model")
    ap.add_argument("--batch", type=int, default=32, help="configs 2/3: utterances per GPU (weak) or in total (strong)")
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--no-pack", action="store_true", help="A/B: ragged batches keep the padded row layout (no host lengths handed to the encoder)")
    ap.add_argument("--ragged", action="store_true", help="configs 2/3: utterance lengths linspace(seconds/2, seconds, batch) instead of equal "
                    "(SURVEY 8d's second run); the CPU reference leg then decodes the batch whole")
    ap.add_argument("--utts-per-gpu", type=int, default=128, help="config 4, weak scaling")
    ap.add_argument("--longform-seconds", type=int, default=3600, help="config 5")
    ap.add_argument("--fr-batch", type=int, default=16, help="config 5: chunks per batch (the reference's fr_batch_size default)")
    ap.add_argument("--layers", type=int, default=-1, help="debug only: fewer layers INVALIDATES the number")
    ap.add_argument("--cpu-utts", type=int, default=-1, help="utterances in the CPU reference leg (0 = skip; default: 32 -- all of configs 2 / 3, "
                    "the first batch of config 4, the first two batches of config 5)")
    ap.add_argument("--no-port-leg", action="store_true", help="config 2: do not time oracle/gigaam_oracle.py beside the reference")
    ap.add_argument("--rnnt-blank-bias", type=float, default=None,
                    help="RNN-T models: blank bias of the synthetic joint (default: the blank-dominant value of tests/golden/fullsize_meta.json)")
    ap.add_argument("--rnnt-overlap", type=int, default=1, help="RNN-T models: 1 = the decode of batch n runs on a side stream beside the "
                    "encoder of batch n+1 (small clusters; configs 3 / 4), 0 = in front of it on the launch stream (full-size clusters)")
    ap.add_argument("--rnnt-side-cus", type=int, default=0, help="compute units an overlapped RNN-T decode may hold (cluster size = this / utterance "
                    "slots); 0 = the engine's choice (96 for a char vocabulary, 160 for a SentencePiece one)")
    ap.add_argument("--pre-streams", type=int, default=0, help="debug: take this many streams from torch's pool before the model exists, as an "
                    "application with streams of its own would (which hardware queue a later stream shares depends on it: tools/queue_probe.py)")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event timing")
    ap.add_argument("--no-power", action="store_true", help="skip the board power / shader clock sampling leg")
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the exact-fp32 re-timing (roofline_f32_exact)")
    ap.add_argument("--no-h2d-leg", action="store_true", help="skip the host-memory re-timing (h2d_ms)")
    ap.add_argument("--no-f16-leg", action="store_true", help="skip the opt-in one-term fp16 speed-mode re-timing (roofline_f16_fast)")
    ap.add_argument("--gemm", default="f16x3", choices=["f16x3", "f32", "f16"],
                    help="dense-contraction arithmetic: split-fp16 MFMA (fp32-equivalent, default), exact fp32 MFMA, or the opt-in "
                         "one-term fp16 speed mode (the reference's GPU autocast contract; its line is marked, never the headline)")
    ap.add_argument("--launch-selftest", action="store_true", help="CPU-only check of the rank plumbing (gloo); no GPU work")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="rehearsal on a box with fewer GPUs than ranks: ranks share devices (rank %% device_count) and "
                         "torch.distributed runs on gloo (RCCL refuses two ranks on one device); the line is marked INVALID")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver's plain `python bench.py --gpus N`: no launcher set the rank environment, so this process becomes it
        if not args.launch_selftest:
            have = torch.cuda.device_count() if torch.cuda.is_available() else 0
            if have < args.gpus and not args.oversubscribe:
                raise SystemExit(f"bench.py --gpus {args.gpus}: {have} GPU(s) visible on this box; one process per GPU needs "
                                 f"{args.gpus} (a rehearsal with ranks sharing devices: --oversubscribe)")
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_ranks = int(os.environ.get("WORLD_SIZE", "1"))
    if n_ranks != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={n_ranks}: launch {args.gpus} ranks "
                         f"(torch.distributed.run --nproc-per-node {args.gpus}) or drop the launcher and let bench.py start them")
    if args.launch_selftest:
        return launch_selftest(rank, n_ranks, args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (no CPU fallback)")
    # host side of the timed region: a handful of small CPU tensor ops.  Left at the default (one OpenMP thread per
    # core) they spin on every core of the box and, under a CPU quota, get the launching thread throttled mid-batch.
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev and not args.oversubscribe:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {n_dev} GPU(s) visible (one process per GPU; --oversubscribe "
                         "for a shared-device rehearsal)")
    shared = args.oversubscribe and n_ranks > n_dev
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    if n_ranks > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            dist.init_process_group("gloo", rank=rank, world_size=n_ranks)
            args.gather = "torch"
        else:
            dist.init_process_group("nccl", rank=rank, world_size=n_ranks, device_id=torch.device("cuda", dev_index))
    dev = torch.device("cuda", dev_index)

    if args.comm_selftest:
        # tools/scale8.sh, first step on an 8-GPU node: the communicator behind the C ABI (gam_comm_create = ncclCommInitRank through
        # dlopen, never yet run with a world > 1) and ONE gam_gather_ids of a tiny buffer, before any model is built or anything is timed --
        # a broken communicator then costs a minute, not the session.  Every rank checks what it received.
        t0 = time.perf_counter()
        gather, gather_name = make_gather(args.gather, rank, n_ranks, dev, strict=True)
        rows, width = 3, 5
        index = torch.arange(rows, dtype=torch.int32, device=dev) + rank * rows
        counts = torch.full((rows,), rank + 1, dtype=torch.int32, device=dev)
        ids = (torch.arange(rows * width, dtype=torch.int32, device=dev).reshape(rows, width) + 1000 * rank)
        frames = ids + 7
        gi, gc, gids, gfr = gather(index, counts, ids, frames)
        torch.cuda.synchronize()
        ok = (gi.cpu().tolist() == list(range(n_ranks * rows)) and gc.cpu().tolist() == [r + 1 for r in range(n_ranks) for _ in range(rows)]
              and all(int(gids[r * rows, 0]) == 1000 * r and int(gfr[r * rows + rows - 1, width - 1]) == 1000 * r + rows * width - 1 + 7 for r in range(n_ranks)))
        flag = torch.tensor([1 if ok else 0], device=dev)
        if n_ranks > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if rank == 0:
            emit({"comm_selftest": "ok" if int(flag.item()) == 1 else "FAILED", "n_gpus": n_ranks, "gather_path": gather_name,
                  "rows_exchanged": n_ranks * rows, "seconds": round(time.perf_counter() - t0, 2)}, n_ranks)
        elif n_ranks > 1:
            dist.destroy_process_group()
        if int(flag.item()) != 1:
            raise SystemExit(f"[bench] rank {rank}: --comm-selftest: the gathered buffers are wrong")
        return

    import gigaam_amd
    from gigaam_amd import synth, workloads

    _pre = []
    for _ in range(args.pre_streams):
        _pre.append(torch.cuda.Stream(dev))
        with torch.cuda.stream(_pre[-1]):
            torch.zeros(8, device=dev).add_(1)

    cfgno = args.config
    model_name = args.model or {1: "v2_ctc", 2: "v2_ctc", 3: "v2_rnnt", 4: "v3_e2e_rnnt", 5: "v2_ctc"}[cfgno]
    over = {} if args.layers < 0 else {"n_layers": args.layers}
    is_rnnt = model_name.endswith("rnnt")
    bias = rnnt_bias_for(model_name, args.rnnt_blank_bias) if is_rnnt else None
    ckpt = synth.make_checkpoint(model_name, seed=0, rnnt_blank_bias=bias, **over)
    model = gigaam_amd.model_from_checkpoint(ckpt, dev)
    eng = model.encoder.engine
    eng.set_gemm_mode(args.gemm)
    max_sym = ckpt["cfg"]["decoding"].get("max_symbols_per_step", 10)
    gather, gather_name = make_gather(args.gather, rank, n_ranks, dev, strict=args.strict_gather)
    scaling = "strong" if cfgno == 5 else args.scaling

    rnnt_overlap = is_rnnt and args.rnnt_overlap != 0

    host_lens = {}      # id(device length tensor) -> (the tensor, its feature lengths as host integers): what the caller of a real
                        # transcribe() has anyway (file lengths) -- a ragged batch then runs on its valid frames only (engine.encode)

    def note_host(wlen_dev, samples):
        if not args.no_pack:
            host_lens[id(wlen_dev)] = (wlen_dev, eng.host_feat_lengths(samples))

    def decode_dev(wav, wlen, overlap=False, host=None):
        """frontend + encoder + greedy decode, launched, no host sync.  ``overlap`` (RNN-T, another batch follows before this one is
        collected): the decode goes to the engine's side stream with small clusters, beside the next batch's encoder.  ``host``: the
        batch's sample counts on the host (else what note_host registered for this length tensor)."""
        feat, flen = eng.frontend(wav, wlen)
        hfl = None
        if not args.no_pack:
            hfl = eng.host_feat_lengths(host) if host is not None else host_lens.get(id(wlen), (None, None))[1]
        enc, elen = eng.encode(feat, flen, host_lengths=hfl)
        if is_rnnt:
            return eng.rnnt_greedy(enc, elen, max_sym, overlap=overlap and rnnt_overlap, side_cus=args.rnnt_side_cus)
        return eng.ctc_greedy(enc, elen)

    drain = lambda: None   # noqa: E731  (pipelined steps: collect what is still in flight)

    # ---- the step of each configuration; `audio_s` = audio seconds ALL ranks process per step
    cpu_sample = None       # ([(wav, wlen)] host batches exactly as the GPU decodes them, their global indices) for the CPU leg
    if cfgno in (1, 2, 3):
        if cfgno == 1:
            n_global, seconds = 1, 5.0
            g0, g1 = (0, 1) if rank == 0 else (0, 0)
        else:
            n_global = args.batch * (n_ranks if scaling == "weak" else 1)
            seconds = args.seconds
            g0, g1 = shard_range(n_global, rank, n_ranks)
        wav_h, wlen_h = workloads.config2_batch(max(1, g1 - g0), seconds, first=g0) if cfgno != 1 else workloads.config1_clip()
        ragged_global = None
        if args.ragged and cfgno != 1:
            # SURVEY 8d's second run of config 2 / 3: lengths linspace(10 s, 20 s, batch) -- masks, ragged frame counts, padded tails
            pat = workloads.config2_ragged_lengths(args.batch, 0.5 * seconds, seconds)
            ragged_global = pat * n_ranks if scaling == "weak" else pat
            mine_l = ragged_global[g0:g1] or [int(seconds * 16000)]
            wav_h, wlen_h = synth.synth_audio(len(mine_l), seconds, seed=1000, index0=g0, lengths=mine_l)
        wav, wlen = wav_h.to(dev), wlen_h.to(dev)          # resident in HBM before the timed region
        note_host(wlen, wlen_h)
        rows = max(shard_range(n_global, r, n_ranks)[1] - shard_range(n_global, r, n_ranks)[0] for r in range(n_ranks))
        audio_s = seconds * n_global if ragged_global is None else sum(ragged_global) / 16000.0
        idx_dev = torch.full((rows,), -1, dtype=torch.int32, device=dev)
        idx_dev[: g1 - g0] = torch.arange(g0, g1, dtype=torch.int32, device=dev)
        cpu_sample = ([(wav_h, wlen_h)], list(range(g0, g1)))

        def step():
            dec = None
            if g1 > g0:
                dec = decode_dev(wav, wlen)
                ids, frames, counts = dec[0], dec[1], dec[2]
            else:   # strong scaling with more ranks than utterances: this rank only takes part in the exchange
                cap0 = eng.enc_frames(eng.feat_frames(int(seconds * 16000))) * (max_sym if is_rnnt else 1)
                ids = frames = torch.zeros((0, cap0), dtype=torch.int32, device=dev)
                counts = torch.zeros((0,), dtype=torch.int32, device=dev)
            if n_ranks > 1:
                # this rank's range flag rides in the exchange as one extra row (shard.append_flag_row): the padding, the
                # gather and the row selection below all drop the hidden tail word of `counts`, and a separate
                # eng.range_flag() here would read a flag the decode call has already consumed
                gi, gc, gids, gfr = gather(*shard.append_flag_row(idx_dev, counts, ids, frames, shard.range_flag_of(dec), rows))
                dec, _, flag = shard.collect_gathered(gi, gc, gids, gfr)
                if flag:
                    raise RuntimeError("split-fp16 range flag set on some rank during a bench step")
                return dec
            return ragged_host(dec)        # the decoded ids (+ the range flag at N = 1) end every step on the host

        if rnnt_overlap and n_ranks == 1 and cfgno != 1:
            # RNN-T (config 3), one rank: the same launch-n / collect-n-1 pipeline configs 4 / 5 and transcribe_longform use,
            # ACROSS steps -- batch n's greedy loop (latency-bound, side stream, small clusters) runs beside batch n+1's
            # encoder, and the ids of batch n reach the host while n+1 is on the GPU.  K steps = K batches launched AND
            # collected between the two barriers (drain() collects the last one inside the timed region).
            in_flight = []
            def step():   # noqa: F811
                in_flight.append(decode_dev(wav, wlen, overlap=True))
                return ragged_host(in_flight.pop(0)) if len(in_flight) > 1 else None

            def drain():  # noqa: F811
                res = None
                while in_flight:
                    res = ragged_host(in_flight.pop(0))
                return res
        workload = (f"{model_name} (16-layer Conformer, random-init weights), {n_global} x "
                    f"{'linspace(%g s, %g s) RAGGED' % (0.5 * seconds, seconds) if ragged_global else '%g s' % seconds} 16 kHz utterances "
                    f"({'%d per GPU' % args.batch if scaling == 'weak' else 'global batch split over the ranks'}), frontend + encoder + "
                    f"{'RNN-T' if is_rnnt else 'CTC'} greedy + ids to host, final gather of ids")
    elif cfgno == 4:
        n_utts = args.utts_per_gpu * n_ranks if scaling == "weak" else 1024
        mine = shard.deal((n_utts + 31) // 32, rank, n_ranks)
        host_batches = workloads.config4_batches(n_utts, 32, only_batches=set(range((n_utts + 31) // 32)) if n_ranks == 1 else None)
        batches = [(w.to(dev) if j in mine else w, l.to(dev) if j in mine else l, g) for j, (w, l, g) in enumerate(host_batches)]
        for j, (_, l_h, _) in enumerate(host_batches):
            if j in mine:
                note_host(batches[j][1], l_h)
        audio_s = float(sum(int(l.sum()) for _, l, _ in host_batches)) / 16000.0
        cap = eng.enc_frames(eng.feat_frames(20 * 16000)) * max_sym
        tok = model.decoding.tokenizer
        if rank == 0 and host_batches:
            cpu_sample = ([(host_batches[mine[0]][0], host_batches[mine[0]][1])], list(host_batches[mine[0]][2]))

        def step():
            # batch n is launched before batch n-1's ids are copied back (shard.run_sharded, collect=)
            res = shard.run_sharded(batches, decode_dev, rank, n_ranks, gather, cap, my_batches=mine, collect=ragged_host, overlap_kw=True)
            return res if rank != 0 else [(i, f, tok.decode(i)) for i, f in res]     # detokenised like the package API
        workload = (f"{model_name} (V = 1025), {n_utts} utterances with durations U(5 s, 20 s) sorted into 32-utterance batches "
                    f"dealt to {n_ranks} rank(s) ({len(mine)} batches on rank 0), frontend + encoder + RNN-T greedy + ids to host + "
                    "gather + detokenise")
    else:
        from gigaam_amd.feeder import BatchFeeder
        segs, bounds = workloads.config5_segments(args.longform_seconds)
        fr_bs = args.fr_batch
        # Who decodes what: CHUNKS (not batches) are dealt by duration, longest first, each to the rank with the least audio so
        # far (shard.lpt_deal), and every rank cuts its own share -- already sorted by length -- into batches of fr_batch_size.
        # Every rank gets the same audio seconds (194 chunks on 8 ranks: max / min = 1.02; dealing whole file-order batches
        # round-robin gave 2,2,2,2,2,1,1,1 batches = a 6.5x ceiling) and every batch holds chunks of
        # neighbouring lengths.  The file order of the reference's loop (gigaam/model.py:219-258) is restored in the result.
        costs = [int(x.shape[0]) for x in segs]
        all_rank_batches = shard.rank_batches(costs, n_ranks, fr_bs)
        my_batches = all_rank_batches[rank]
        my_idx = [i for b in my_batches for i in b]
        my_segs = [segs[i] for i in my_idx]
        per_rank = max(sum(len(b) for b in rb) for rb in all_rank_batches)
        rank_audio = [sum(costs[i] for b in rb for i in b) / 16000.0 for rb in all_rank_batches]
        cap = eng.enc_frames(eng.feat_frames(30 * 16000 + 160))
        audio_s = float(args.longform_seconds)
        tok = model.decoding.tokenizer

        feeder = BatchFeeder(my_segs, fr_bs, dev) if my_segs else []       # pinned staging buffers: allocated once
        last5 = {}
        if rank == 0:
            # CPU leg: rank 0's first two batches (2 x fr_batch_size chunks; LPT deals the file's longest chunk first, to rank 0),
            # each collated on its own -- the reference's features of an utterance's last frames depend on what follows it in
            # its row (zero padding inside a batch, reflect padding at the end of the longest row: torchaudio center=True pads
            # the TENSOR), so the reference must see each chunk exactly as the GPU's batch held it
            from gigaam_amd.feeder import collate
            b5, g5 = [], []
            for rows5 in my_batches[:2]:
                b5.append(collate([segs[i] for i in rows5]))
                g5 += list(rows5)
            cpu_sample = (b5, g5)

        trace = os.environ.get("GAM_BENCH_TRACE")   # debug: host timestamps per batch (ms since the step began)

        def step():
            rows, pending = [], None
            t_s, marks = time.perf_counter(), []
            for wav_b, len_b in feeder:                                       # pinned, double-buffered H2D
                t_a = time.perf_counter()
                out_b = decode_dev(wav_b, len_b, host=feeder.host_lengths)    # launched; collected one batch later
                t_b = time.perf_counter()
                if pending is not None:
                    rows += [(my_idx[len(rows) + k], i, f) for k, (i, f) in enumerate(ragged_host(pending))]
                pending = out_b
                if trace:
                    marks.append((round((t_a - t_s) * 1e3, 1), round((t_b - t_a) * 1e3, 1), round((time.perf_counter() - t_b) * 1e3, 1), tuple(wav_b.shape)))
            if trace:
                print("[trace] (staged_at, launch_ms, collect_prev_ms, shape):", marks, file=sys.stderr)
            if pending is not None:
                rows += [(my_idx[len(rows) + k], i, f) for k, (i, f) in enumerate(ragged_host(pending))]
            res = shard.unpack_results(*gather(*shard.pack_results(rows, per_rank, cap)), len(segs))
            last5["res"] = res
            return res if rank != 0 else [(tok.decode(i), bounds[k]) for k, (i, f) in enumerate(res)]
        workload = (f"{model_name} longform: {args.longform_seconds} s of audio -> {len(segs)} chunks (reference packer 22/15/30/0.2 s) -> "
                    f"chunks dealt by duration (LPT) to {n_ranks} rank(s), each rank's share in length-sorted batches of {fr_bs} "
                    f"({len(my_batches)} batches on rank 0; audio seconds per rank max/min = {max(rank_audio) / max(1e-9, min(rank_audio)):.3f}), "
                    "streamed from host memory through the pinned double-buffered feeder, CTC greedy, gather, detokenise, file order restored")
        dealing = {"audio_seconds_per_rank": [round(a, 1) for a in rank_audio], "chunks_per_rank": [sum(len(b) for b in rb) for rb in all_rank_batches],
                   "dealing_bound_speedup": round(sum(rank_audio) / max(rank_audio), 3)}

    # ---- timing: W warm-up steps, then EXACTLY K steps between barrier + synchronize pairs; max over ranks
    # Per-launch HIP-event pairs around the GEMM family INSIDE the timed region (the roofline's launch durations), on every
    # PROF_EVERY-th timed step: an event pair costs ~3 us on this runtime (it drains the queue between kernels), 0.9 ms of a
    # fully instrumented 33 ms step and 1 ms of an 8 ms one -- sampling keeps the measurement from moving what it measures (r06: every 10th
    # step instead of every 4th -- timed_region.host_ms_per_step showed the instrumented steps 1.0 ms long; 2 x 130 launches are sample enough).
    PROF_EVERY = 10
    n_prof_steps = (args.steps + PROF_EVERY - 1) // PROF_EVERY

    timed_diag = {}

    def timed(k_steps, profile):
        barrier_sync()
        if profile:
            eng.profile_enable(0)      # reset the collectors
        allocs0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
        t0 = time.perf_counter()
        out_ = None
        marks = []
        for i in range(k_steps):
            if profile:
                eng.profile_level(2 if i % PROF_EVERY == 0 else 0)
            o = step()
            marks.append(time.perf_counter())
            out_ = o if o is not None else out_
        o = drain()
        out_ = o if o is not None else out_
        barrier_sync()
        t1 = time.perf_counter()
        if profile:
            eng.profile_level(0)
        # diagnostics of the timed region (not part of the contract): hipMalloc calls of torch's caching allocator inside it (each one
        # synchronises the device: a pipelined step that still grows its pool is not in steady state) and the host-side time of every step
        timed_diag.update({"device_allocs": int(torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - allocs0),
                           "host_ms_per_step": [round((b - a) * 1e3, 2) for a, b in zip([t0] + marks[:-1], marks)],
                           "drain_ms": round((t1 - marks[-1]) * 1e3, 2) if marks else None})
        return max_over_ranks(t1 - t0), out_

    for _ in range(args.warmup):
        step()
    drain()
    do_prof = not args.no_profile
    dt, out = timed(args.steps, do_prof)
    timed_region = dict(timed_diag)
    prof = prof_all = None
    if do_prof:
        prof = eng.profile_read()
        eng.profile_enable(1)      # one extra, untimed step for the per-class breakdown
        step()
        drain()
        torch.cuda.synchronize()
        prof_all = eng.profile_read()
        eng.profile_enable(0)

    # board power and shader clock over ~2 s of untimed steps (one rank: the hwmon files are per board)
    power = None
    if n_ranks == 1 and not args.no_power and cfgno in (2, 3):
        n_pw = max(5, int(2.0 / max(1e-3, dt / args.steps)))
        with PowerSampler(dev.index or 0) as ps:
            for _ in range(n_pw):
                step()
            drain()
            torch.cuda.synchronize()
        power = ps.summary()
        power["steps_sampled"] = n_pw

    # host-memory leg (config 2, one rank): the same step with the waveform starting in HOST memory -- never `value` (the
    # contract times inputs resident in HBM), reported next to it.  "serial": one pinned -> device copy on the launch stream in
    # front of every step; "feeder": the product's longform feeding path (feeder.BatchFeeder: the batch is assembled in a
    # pinned staging buffer and copied on a side stream while the previous step's kernels run).
    # pipelined leg (configs 2 / 1 ragged or not, one rank, CTC): the same K batches with the ids of batch n collected AFTER batch n + 1 is
    # launched -- what transcribe_longform / shard.run_sharded do and what config 3's step already is; the D2H copy and the host-side list
    # building then run under the next batch's kernels.  Reported beside `value` (whose step ends with its own ids on the host), never as it.
    pipe_leg = None
    if n_ranks == 1 and cfgno == 2 and not is_rnnt and not args.no_h2d_leg:
        k_p = max(3, min(args.steps, 10))
        ragged_host(decode_dev(wav, wlen))
        barrier_sync()
        t0 = time.perf_counter()
        pend = None
        for _ in range(k_p):
            nxt = decode_dev(wav, wlen)
            if pend is not None:
                ragged_host(pend)
            pend = nxt
        ragged_host(pend)
        barrier_sync()
        t_pipe = (time.perf_counter() - t0) / k_p
        pipe_leg = {"ms_per_step": round(t_pipe * 1e3, 3), "value": round(audio_s / t_pipe, 1), "steps": k_p,
                    "note": "ids of batch n collected after batch n + 1 is launched (the product's launch_batch / collect_batch pipeline); "
                            "informational -- `value` ends every step with its own ids on the host"}

    h2d_leg = None
    if n_ranks == 1 and cfgno == 2 and not args.no_h2d_leg:
        from gigaam_amd.feeder import BatchFeeder
        wav_pin, len_pin = wav_h.pin_memory(), wlen_h.pin_memory()
        k_h = max(3, min(args.steps, 10))

        def step_serial():
            w = torch.empty(wav_pin.shape, dtype=wav_pin.dtype, device=dev)
            w.copy_(wav_pin, non_blocking=True)
            return ragged_host(decode_dev(w, len_pin.to(dev, non_blocking=True)))
        step_serial()
        barrier_sync()
        t0 = time.perf_counter()
        for _ in range(k_h):
            step_serial()
        barrier_sync()
        t_serial = (time.perf_counter() - t0) / k_h
        utt_h = [wav_h[i] for i in range(wav_h.shape[0])]

        def run_feeder(fd):
            pending, n_done = None, 0
            for wb, lb in fd:
                out_b = decode_dev(wb, lb, host=fd.host_lengths)
                if pending is not None:
                    ragged_host(pending); n_done += 1
                pending = out_b
            ragged_host(pending)
            return n_done + 1
        run_feeder(BatchFeeder(utt_h * 2, len(utt_h), dev))
        fd = BatchFeeder(utt_h * k_h, len(utt_h), dev)      # (its two pinned staging buffers are allocated here, outside the timed region)
        barrier_sync()
        t0 = time.perf_counter()
        n_b = run_feeder(fd)
        barrier_sync()
        t_feed = (time.perf_counter() - t0) / n_b
        h2d_leg = {"ms_per_step_serial_copy": round(t_serial * 1e3, 3), "ms_per_step_feeder": round(t_feed * 1e3, 3), "steps": k_h,
                   "feeder_host_collate_ms_per_batch": round(fd.collate_seconds / max(1, fd.batches_staged) * 1e3, 3),
                   "h2d_bytes_per_step": int(wav_h.numel() * 4),
                   "note": "waveform starts in host memory; serial = pinned->device copy on the launch stream before every step; "
                           "feeder = gigaam_amd.feeder.BatchFeeder (batch collated into a pinned staging buffer, side-stream copy "
                           "under the previous step, ids of step n collected after step n+1 is launched)"}

    # exact-fp32 leg: the same steps with the dense contractions on v_mfma_f32_32x32x2_f32 (the reference's arithmetic)
    f32_leg = None
    if args.gemm == "f16x3" and not args.no_f32_leg and cfgno in (2, 3):
        eng.set_gemm_mode("f32")
        for _ in range(2):
            step()
        drain()
        dt32, out32 = timed(args.steps, do_prof)
        p32 = eng.profile_read() if do_prof else None
        eng.profile_enable(0)
        eng.set_gemm_mode("f16x3")
        f32_leg = (dt32, out32, p32)

    # opt-in speed mode leg: the same steps with ONE fp16 MFMA per product (GAM_GEMM_F16: the reference's own GPU contract, fp16
    # autocast) -- reported beside the headline with how the decoded ids compare and how wide the default mode's CTC margins are
    f16_leg = None
    if args.gemm == "f16x3" and not args.no_f16_leg and n_ranks == 1 and cfgno in (2, 3):
        eng.set_gemm_mode("f16")
        for _ in range(2):
            step()
        drain()
        dt16, out16 = timed(args.steps, do_prof)
        p16 = eng.profile_read() if do_prof else None
        eng.profile_enable(0)
        margins = None
        if not is_rnnt:
            with torch.no_grad():
                enc16, elen16 = eng.encode(*eng.frontend(wav, wlen))
                lp16 = eng.ctc_head(enc16)
                eng.set_gemm_mode("f16x3")
                enc3, elen3 = eng.encode(*eng.frontend(wav, wlen))
                lp3 = eng.ctc_head(enc3)
                valid = torch.arange(lp3.shape[1], device=dev)[None, :] < elen3[:, None]
                top2 = lp3.topk(2, dim=-1).values
                mg = (top2[..., 0] - top2[..., 1])[valid]
                edges = [0.0, 1e-3, 1e-2, 1e-1, 1.0, float("inf")]
                hist = [int(((mg >= lo) & (mg < hi)).sum()) for lo, hi in zip(edges[:-1], edges[1:])]
                flips = int(((lp16.argmax(-1) != lp3.argmax(-1)) & valid).sum())
                margins = {"default_mode_top1_top2_margin_histogram": dict(zip(["<1e-3", "1e-3..1e-2", "1e-2..1e-1", "1e-1..1", ">=1"], hist)),
                           "frames": int(valid.sum()), "frames_whose_argmax_differs_in_f16": flips,
                           "encoder_max_abs_diff_vs_default": round(float(((enc16 - enc3).abs() * valid[:, None, :]).max()), 5)}
        eng.set_gemm_mode("f16x3")
        f16_leg = (dt16, out16, p16, margins)

    if rank != 0:
        if n_ranks > 1:
            dist.destroy_process_group()
        return

    ms_step = dt / args.steps * 1e3
    fam = ("gemm", "conv2")   # one kernel family: the GEMMs incl. the implicit-GEMM stem conv

    def family(p, peak):
        flop = sum(p[k]["work"] for k in fam)
        ms = sum(p[k]["ms"] for k in fam)
        n = sum(p[k]["launches"] for k in fam)
        ach = flop / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        return flop, ms, n, ach, ach / peak

    metric = {1: "RTFx v2_ctc single 5 s clip", 2: f"RTFx {model_name} batch{args.batch}x{args.seconds:g}s (log-mel + encoder + greedy decode)",
              3: f"RTFx {model_name} batch{args.batch}x{args.seconds:g}s (log-mel + encoder + RNN-T greedy decode)",
              4: "RTFx v3_e2e_rnnt utterance set sharded over the ranks", 5: "RTFx v2_ctc longform streamed over the ranks"}[cfgno]
    n_units = {1: 1, 2: args.batch, 3: args.batch}.get(cfgno)
    line = {
        "metric": metric, "value": round(audio_s * args.steps / dt, 1), "unit": "audio-sec/wall-sec", "n_gpus": n_ranks,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": {"f32": "f32", "f16": "f16 products, f32 accumulate (OPT-IN speed mode: narrower than the CPU reference)"}.get(
            args.gemm, "f32 (GEMMs: 3-term split on fp16 MFMA, fp32 accumulate)"),
        "data": "synthetic",
        "config": {"workload": workload, "baseline_config": cfgno, "audio_seconds_per_step": round(audio_s, 1),
                   "parallelism": f"dp{n_ranks} (utterance shards, replicated weights, one exchange: {gather_name})"},
    }
    try:
        r_run, r_pad = eng.last_encode_rows()
        line["config"]["token_rows_last_batch"] = {"run": r_run, "padded_layout": r_pad, "packed": bool(r_run < r_pad)}
    except Exception:   # noqa: BLE001
        pass
    line["timed_region"] = timed_region
    line["gather_path"] = gather_name        # (top level: a silent fall-back to torch.distributed on the 8-GPU node must be visible)
    if cfgno == 5:
        line["config"]["dealing"] = dealing
    if n_units:
        line["config"]["global_batch"] = n_units * (n_ranks if scaling == "weak" else 1)
        line["encoder_ms_per_utt"] = round(ms_step / max(1, (g1 - g0)), 4)
    if cfgno in (1, 2, 3):
        line["tokens_decoded_per_step"] = int(sum(len(i) for i, _ in out))
        decoded_mine = out[g0:g1] if n_ranks > 1 else out
    else:
        line["tokens_decoded_per_step"] = int(sum(len(r[0]) for r in out))
        decoded_mine = None
    if is_rnnt:
        line["rnnt_blank_bias"] = bias
    if args.layers >= 0:
        line["INVALID"] = "debug run with --layers"
    if args.gemm == "f16":
        line["INVALID"] = "opt-in fp16 speed mode (--gemm f16): narrower arithmetic than the fp32 CPU reference -- not a headline number"
    if shared:
        line["INVALID"] = f"rehearsal: {n_ranks} ranks share {n_dev} GPU(s) (--oversubscribe), exchange over gloo -- not a scaling measurement"
    if prof is not None:
        if args.gemm == "f32":
            kern, peak, peak_note = "gam_gemm_f32_kernel (v_mfma_f32_32x32x2_f32; plain + implicit-GEMM conv)", FP32_MFMA_PEAK_TFLOPS, \
                "fp32 dense MFMA peak"
        else:
            # every algorithmic FLOP costs three fp16 MFMA FLOPs (hi.hi + hi.lo + lo.hi): the ceiling of
            # this arithmetic is a third of the fp16 dense peak
            kern = ("gam_gemm_sp_kernel (LDS-DMA, sp32 operands, epilogue straight from the swapped-operand accumulators; split-K "
                    "slices for small grids) -- 3x v_mfma_f32_32x32x16_f16 per product; plain + implicit-GEMM conv")
            peak, peak_note = F16_MFMA_PEAK_TFLOPS / 3.0, "fp16 dense MFMA peak (2500) / 3 issued MFMA FLOP per algorithmic FLOP"
            if args.gemm == "f16":
                kern = ("gam_gemm_sp_kernel<.., H16> (format-2 operands: plain fp16 rows written by the producers, the kernel told half "
                        "the reduction length; two v_mfma_f32_32x32x16_f16 per 32x32x32 block = one MFMA per product)")
                peak, peak_note = F16_MFMA_PEAK_TFLOPS, "fp16 dense MFMA peak"
        flop, ms, n, ach, frac = family(prof, peak)
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", f"pmc_traffic_{args.gemm}.json")
        if cfgno == 2 and args.batch == 32 and args.seconds == 20.0 and os.path.exists(tpath):
            tj = json.load(open(tpath))       # committed rocprofv3 PMC passes of this same command (not re-measured in this run)
            traffic, traffic_src = round(tj["traffic_bytes_per_launch"]), os.path.relpath(tpath, ROOT) + " (committed rocprofv3 --pmc passes of this command)"
        alg_bytes = sum(prof[k].get("bytes", 0.0) for k in fam)
        line["roofline"] = {
            "kernel": kern, "bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(frac, 4), "traffic": traffic, "traffic_source": traffic_src,
            "peak_note": peak_note, "issued_mfma_tflops": round(ach * (1.0 if args.gemm == "f32" else 3.0), 1),
            "algorithmic_bytes_per_launch": round(alg_bytes / max(1, n)),
            "launches_per_step": n // max(1, n_prof_steps), "avg_launch_ms": round(ms / max(1, n), 4),
            "algorithmic_gflop_per_step": round(flop / n_prof_steps / 1e9, 1),
            "share_of_step_time": round(ms / n_prof_steps / ms_step, 3),
            "events": f"per-launch HIP event pairs on every {PROF_EVERY}th timed step ({n_prof_steps} of {args.steps}: steps 0, {PROF_EVERY}, ...)",
        }
        line["kernel_classes_ms_per_step"] = {k: round(v["ms"], 3) for k, v in prof_all.items() if v["launches"]}
        line["kernel_classes_note"] = "HIP-event time per class from one extra untimed step"
        if is_rnnt and cfgno == 3 and "decode" in prof_all:
            # the greedy loop is latency-bound -- report joint evaluations ("steps": one per frame + one per
            # emitted token, decoding.py:162-205) per second and the time per step on the longest utterance's chain
            with torch.no_grad():
                elen = eng.encode(*eng.frontend(wav, wlen))[1].cpu().tolist()
            steps = [int(e) + len(i) for e, (i, _) in zip(elen, decoded_mine)]
            dec_ms = prof_all["decode"]["ms"]
            line["rnnt_decode"] = {"ms_per_batch": round(dec_ms, 3), "joint_steps_per_batch": int(sum(steps)),
                                   "steps_per_s": round(sum(steps) / (dec_ms * 1e-3)),
                                   "us_per_step_longest_utterance": round(dec_ms * 1e3 / max(1, max(steps)), 3),
                                   "overlapped_with_next_batch_encoder": bool(rnnt_overlap and n_ranks == 1),
                                   "note": "768->320 projection GEMM + gam_rnnt_cluster_kernel; the utterances of a batch decode "
                                           "concurrently (one workgroup cluster each), a 16-frame window per hand-off round; when overlapped "
                                           "(side stream, small clusters) ms_per_batch is the decode's own span, mostly hidden under the next encoder"}
        if cfgno in (2, 3):
            whole = FLOP_PER_UTT_20S_V2 * (float(wlen_h.sum()) / 16000.0 / 20.0 if g1 > g0 else 0.0) / (ms_step * 1e-3) / 1e12
            line["whole_path_tflops_per_gpu"] = round(whole, 2)
    if power is not None:
        line["board_power"] = power
        if power.get("avg_sclk_mhz") and "roofline" in line and line["roofline"]["bound"] == "mfma":
            # the peak is quoted at the 2400 MHz boost clock; under this load the board sits at its power cap and the
            # shader clock is held lower -- the fraction of what the matrix cores can issue AT THAT CLOCK is reported next
            # to `frac` (never instead of it)
            r = line["roofline"]
            r["avg_sclk_mhz_under_load"] = power["avg_sclk_mhz"]
            r["frac_at_measured_clock"] = round(r["frac"] * PEAK_SCLK_MHZ / power["avg_sclk_mhz"], 4)
    if pipe_leg is not None:
        line["pipelined"] = pipe_leg
    if h2d_leg is not None:
        line["h2d_ms"] = h2d_leg["ms_per_step_feeder"]
        line["h2d"] = h2d_leg
    if f32_leg is not None:
        dt32, out32, p32 = f32_leg
        leg = {"ms_per_step": round(dt32 / args.steps * 1e3, 3), "value": round(audio_s * args.steps / dt32, 1),
               "unit": "audio-sec/wall-sec", "dtype": "f32 (v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 fmaf chain)",
               "ids_identical_to_default_mode": sum(a == b for a, b in zip(out32, out)), "utterances": len(out)}
        if p32 is not None:
            flop, ms, n, ach, frac = family(p32, FP32_MFMA_PEAK_TFLOPS)
            leg.update(achieved=round(ach, 2), peak=FP32_MFMA_PEAK_TFLOPS, frac=round(frac, 4), launches_per_step=n // max(1, n_prof_steps),
                       avg_launch_ms=round(ms / max(1, n), 4))
        line["roofline_f32_exact"] = leg
    if f16_leg is not None:
        dt16, out16, p16, margins = f16_leg
        leg = {"ms_per_step": round(dt16 / args.steps * 1e3, 3), "value": round(audio_s * args.steps / dt16, 1), "unit": "audio-sec/wall-sec",
               "dtype": "fp16 products (one v_mfma_f32_32x32x16_f16 per product on plain-fp16 format-2 operands), fp32 accumulate; fp32 softmax / LayerNorm / residual",
               "ids_identical_to_default_mode": sum(a == b for a, b in zip(out16, out)), "utterances": len(out),
               "note": "OPT-IN (gam_set_gemm_mode(GAM_GEMM_F16) / GAM_GEMM_MODE=f16 / model.set_arithmetic('f16')): the arithmetic contract of the "
                       "reference's GPU default (fp16 autocast, gigaam/model.py:34-37), narrower than its CPU path -- never the default, never `value`"}
        if p16 is not None:
            flop, ms, n, ach, frac = family(p16, F16_MFMA_PEAK_TFLOPS)
            leg.update(achieved=round(ach, 2), peak=F16_MFMA_PEAK_TFLOPS, frac=round(frac, 4), launches_per_step=n // max(1, n_prof_steps),
                       avg_launch_ms=round(ms / max(1, n), 4))
        if margins is not None:
            leg["ctc_margins"] = margins
        line["roofline_f16_fast"] = leg
    n_cpu = args.cpu_utts if args.cpu_utts >= 0 else 32
    if n_ranks == 1 and n_cpu > 0 and cpu_sample is not None:
        try:
            cb, gidx = cpu_sample
            # at most n_cpu utterances: whole batches, the last one cut by ROWS only (the padded width -- what the reference's
            # reflect / zero padding sees -- stays the batch's own)
            keep, left = [], n_cpu
            for w_h, l_h in cb:
                if left <= 0:
                    break
                k = min(left, int(w_h.shape[0]))
                keep.append((w_h[:k].contiguous(), l_h[:k].contiguous()))
                left -= k
            offs, sel = 0, []
            for (w_h, _), (wk, _) in zip(cb, keep):
                sel += gidx[offs: offs + int(wk.shape[0])]
                offs += int(w_h.shape[0])
            if cfgno in (1, 2, 3):
                gpu_dec = [decoded_mine[g - g0] for g in sel]
            elif cfgno == 4:
                gpu_dec = [(out[g][0], out[g][1]) for g in sel]
            else:
                gpu_dec = [tuple(last5["res"][g]) for g in sel]
            def gpu_probe(i):
                """(GPU log-probs of utterance i of the CPU sample -- per frame for CTC, per joint evaluation for RNN-T, decoded in its own
                batch exactly as the timed step decoded it, same cluster size --, extra fields for the report)."""
                from gigaam_amd.engine import HipEngine
                bi, r = 0, i
                while r >= int(keep[bi][0].shape[0]):
                    r -= int(keep[bi][0].shape[0])
                    bi += 1
                w_d, l_d = keep[bi][0].to(dev), keep[bi][1].to(dev)
                enc_d, elen_d = eng.encode(*eng.frontend(w_d, l_d))
                if not is_rnnt:
                    return eng.ctc_head(enc_d)[r, :int(elen_d[r])].cpu(), {}
                b_, tp_ = int(enc_d.shape[0]), int(enc_d.shape[2])
                cap = tp_ + max(len(g[0]) for g in gpu_dec) + 16
                c_timed = (HipEngine.side_cluster(b_, args.rnnt_side_cus if args.rnnt_side_cus > 0 else (96 if eng.cfg.num_classes <= 64 else 160))
                           if (rnnt_overlap and cfgno != 1) else -1)
                eng.set_rnnt_cluster(c_timed)
                try:
                    d = eng.rnnt_greedy(enc_d, elen_d, max_sym, dump_cap=cap)
                    ids_t = HipEngine.collect(d)[0][r]
                    lp = d[3][r, :int(d[4][r])].cpu()
                finally:
                    eng.set_rnnt_cluster(-1)
                ids_full = HipEngine.collect(eng.rnnt_greedy(enc_d, elen_d, max_sym))[0][r]
                want = cpu_ids_of.get(i)
                return lp, {"cluster_size_timed": c_timed, "ids_identical_at_full_cluster_size": (None if want is None else bool(list(ids_full[0]) == want[0] and list(ids_full[1]) == want[1])),
                            "timed_ids_reproduced_by_probe": bool((list(ids_t[0]), list(ids_t[1])) == (list(gpu_dec[i][0]), list(gpu_dec[i][1])))}
            cpu_ids_of = {}
            line["cpu_baseline"] = cpu_baseline(ckpt, keep, gpu_dec, sweep=(cfgno == 2), equal_lengths=cfgno in (1, 2, 3) and not args.ragged,
                                                port_too=(cfgno == 2 and not args.no_port_leg), gpu_probe=gpu_probe if cfgno in (1, 2, 3, 4) else None,
                                                ref_ids_out=cpu_ids_of)
        except Exception as e:  # the bench line must still print
            import traceback
            line["cpu_baseline"] = {"error": repr(e), "traceback": traceback.format_exc()[-600:]}
    emit(line, n_ranks)


def emit(line, n_ranks):
    """The JSON line must be the LAST thing on stdout: RCCL prints its version banner through C stdio, which is
    block-buffered on a pipe and would otherwise surface after the line, at exit."""
    import ctypes
    if n_ranks > 1:
        dist.destroy_process_group()
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(line, ensure_ascii=False), flush=True)


if __name__ == "__main__":
    main()
