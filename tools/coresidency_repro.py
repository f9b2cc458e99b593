"""r06: the one-switch reproducer of the RNN-T cluster decode's co-residency perturbation.
A dense, near-tie-rich batch (v2_rnnt, 2 layers, blank bias 13.5: up to 10 symbols per frame) is decoded on the engine's side stream while the
launch stream runs the small-tile LDS-DMA GEMM; ids are compared with the decode that had the GPU to itself at the SAME cluster size.
    GAM_RNNT_EXCLUSIVE=0 [GIGAAM_HIP_LIB=...] python tools/coresidency_repro.py <clusters e.g. 2,1> <reps> [neighbour=gemm640|legacy|attention|none]
prints one line per cluster size: decodes that differ / decodes run."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gigaam_amd import synth  # noqa: E402
from gigaam_amd.engine import HipEngine, build_config  # noqa: E402

clusters = [int(c) for c in sys.argv[1].split(",")]
reps = int(sys.argv[2])
beside = sys.argv[3] if len(sys.argv) > 3 else "gemm640"
ck = synth.make_checkpoint("v2_rnnt", seed=1, n_layers=2, rnnt_blank_bias=13.5)
cfg = ck["cfg"]
eng = HipEngine(build_config(cfg["preprocessor"], cfg["encoder"], cfg["head"]), ck["state_dict"], torch.device("cuda:0"))
lens = [int(16000 * (1.0 + 0.37 * ((3 * i + 1) % 11))) for i in range(32)]
wav, wlen = synth.synth_audio(32, max(lens) / 16000.0, seed=301, lengths=lens)
enc, elen = eng.encode(*eng.frontend(wav, wlen))
xa = torch.randn(640, 768, device="cuda")
wa = torch.randn(768, 768, device="cuda") * 0.03
qa = torch.randn(5, 120, 768, device="cuda")
la = torch.tensor([120, 100, 90, 77, 50], device="cuda")
ref_gemm = eng.op_gemm(xa, wa).clone()
tag = f"lib={os.path.basename(os.environ.get('GIGAAM_HIP_LIB', 'libgigaam_hip.so'))} exclusive={os.environ.get('GAM_RNNT_EXCLUSIVE', '1')} dbg={os.environ.get('GAM_RNNT_DBG', '0')} beside={beside}"
for cluster in clusters:
    eng.set_rnnt_cluster(cluster)
    alone = HipEngine.collect(eng.rnnt_greedy(enc, elen, 10))[0]
    alone2 = HipEngine.collect(eng.rnnt_greedy(enc, elen, 10))[0]
    eng.set_rnnt_cluster(-1)
    assert alone == alone2
    os.environ["GAM_DEBUG_SIDE_CLUSTER"] = str(cluster)
    bad = bad_utts = gemm_bad = 0
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        dec = eng.rnnt_greedy(enc, elen, 10, overlap=True)
        if beside in ("gemm640", "legacy"):
            for _ in range(30):
                out = eng.op_gemm(xa, wa)
            gemm_bad += not torch.equal(out, ref_gemm)
        elif beside == "attention":
            for _ in range(30):
                eng.op_attention(qa, qa, qa, la)
        got = HipEngine.collect(dec)[0]
        bad += got != alone
        if got != alone:
            u = [i for i, (x, y) in enumerate(zip(got, alone)) if x != y]
            print(f"MISMATCH cluster {cluster} rep {_} utterances {u}", file=sys.stderr, flush=True)
        bad_utts += sum(a != b for a, b in zip(got, alone))
    t1.record(); torch.cuda.synchronize()
    del os.environ["GAM_DEBUG_SIDE_CLUSTER"]
    print(f"REPRO {tag} cluster={cluster}: {bad} / {reps} decodes differ ({bad_utts} utterances), neighbour GEMM results differing {gemm_bad}, "
          f"{t0.elapsed_time(t1) / reps:.2f} ms per rep", flush=True)
