"""r06: are the ENCODER's kernels bit-stable beside another stream's MFMA work?  (the packed-fp32 op_sel erratum was found in the
RNN-T decode; r05's library also held the unreliable form in gam_convmod_ln_kernel -- the v3 conv module.)

Victim: frontend + 2-layer encoder of a small ragged batch (its LayerNorm / attention / conv-module workgroups fit beside a 4-wave GEMM workgroup
on one CU) on the launch stream.  Aggressor: the library's small-tile LDS-DMA GEMM in a loop on a second stream.  The encoder output is compared
BIT FOR BIT with the run that had the GPU to itself.
    [GIGAAM_HIP_LIB=...] python tools/encoder_coresidency.py <reps> [models=v3_e2e_ctc,v2_ctc,v1_ctc] [aggressor_rows=4096] [aggressor_launches_per_run=60]
prints one ENCCO line per model: runs that differ / runs, largest absolute difference seen."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gigaam_amd import synth  # noqa: E402
from gigaam_amd.engine import HipEngine, build_config  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
models = (sys.argv[2] if len(sys.argv) > 2 else "v3_e2e_ctc,v2_ctc,v1_ctc").split(",")
arows = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
loops = int(sys.argv[4]) if len(sys.argv) > 4 else 60      # aggressor launches per run: long enough to cover the whole victim
dev = torch.device("cuda:0")
lib = os.path.basename(os.environ.get("GIGAAM_HIP_LIB", "libgigaam_hip.so"))
side = torch.cuda.Stream(dev)
for name in models:
    ck = synth.make_checkpoint(name, seed=2, n_layers=2)
    cfg = ck["cfg"]
    eng = HipEngine(build_config(cfg["preprocessor"], cfg["encoder"], cfg.get("head")), ck["state_dict"], dev)
    agg = HipEngine(build_config(cfg["preprocessor"], cfg["encoder"], cfg.get("head")), ck["state_dict"], dev)   # its own handle (workspace)
    lens = [int(16000 * (1.0 + 0.41 * ((5 * i + 2) % 8))) for i in range(8)]
    wav, wlen = synth.synth_audio(8, max(lens) / 16000.0, seed=77, lengths=lens)
    wav, wlen = wav.to(dev), wlen.to(dev)
    xa = torch.randn(arows, 768, device=dev)
    wa = torch.randn(768, 768, device=dev) * 0.03
    ref_g = agg.op_gemm(xa, wa).clone()

    def victim():
        enc, elen = eng.encode(*eng.frontend(wav, wlen))
        return enc

    alone = victim().clone()
    assert torch.equal(alone, victim()), "the encoder is not deterministic even alone"
    torch.cuda.synchronize()
    ndiff = torch.zeros((), dtype=torch.int64, device=dev)
    gdiff = torch.zeros((), dtype=torch.int64, device=dev)
    worst = torch.zeros((), dtype=torch.float32, device=dev)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(loops):
                out = agg.op_gemm(xa, wa)
            gdiff += (out != ref_g).any().long()
        got = victim()
        d = (got - alone).abs().max()
        ndiff += (got != alone).any().long()
        worst = torch.maximum(worst, d)
        torch.cuda.current_stream(dev).wait_stream(side)
    t1.record()
    torch.cuda.synchronize()
    print(f"ENCCO lib={lib} model={name} aggressor=gemm {arows}x768x768 x{loops} per run: {int(ndiff)} / {reps} encoder outputs differ from the solo run "
          f"(largest |difference| {float(worst):.3g}); aggressor results differing {int(gdiff)}; {t0.elapsed_time(t1) / reps:.2f} ms per run", flush=True)
