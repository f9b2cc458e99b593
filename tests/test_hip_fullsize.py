"""GPU: BASELINE.json full-size configuration (16 layers, 32 x 20 s) through
size-independent properties -- the CPU oracle is too slow to run here in seconds."""
import pytest
import torch

from common import ragged_from_device

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full_model():
    from gigaam_amd import synth
    from gigaam_amd.engine import HipEngine, build_config
    ck = synth.make_checkpoint("v2_ctc", seed=0)
    cfg = ck["cfg"]
    eng = HipEngine(build_config(cfg["preprocessor"], cfg["encoder"], cfg["head"]), ck["state_dict"], torch.device("cuda:0"))
    lens = [320000 - 5000 * i for i in range(8)]
    wav, wlen = synth.synth_audio(8, 20.0, seed=3, lengths=lens)
    return eng, wav, wlen


def test_fullsize_shapes_determinism_and_padding_invariance(full_model):
    eng, wav, wlen = full_model
    feat, flen = eng.frontend(wav, wlen)
    assert feat.shape == (8, 64, 2001) and flen.cpu().tolist() == [int(l) // 160 + 1 for l in wlen]
    enc, elen = eng.encode(feat, flen)
    assert enc.shape == (8, 768, 501) and elen.dtype == torch.int32 and bool(torch.isfinite(enc).all())
    out = ragged_from_device(*eng.ctc_greedy(enc, elen))
    # idempotence / determinism: a second pass is bit-identical
    enc2, _ = eng.encode(*eng.frontend(wav, wlen))
    assert torch.equal(enc, enc2)
    assert ragged_from_device(*eng.ctc_greedy(enc2, elen)) == out
    # CTC structural invariants (decoding.py:78-82)
    for (ids, frames), n in zip(out, elen.cpu().tolist()):
        assert len(ids) == len(frames) and all(0 <= i < 33 for i in ids)
        assert frames == sorted(set(frames)) and (not frames or frames[-1] < n)
        assert all(a != b or fb > fa + 1 for a, b, fa, fb in zip(ids, ids[1:], frames, frames[1:]))
    assert sum(len(i) for i, _ in out) > 100  # non-degenerate decode
    # batch composition / padding invariance: utterance 5's features alone == inside the batch
    tf = int(flen[5])
    e1, el1 = eng.encode(feat[5:6, :, :tf].contiguous(), flen[5:6])
    t = int(el1[0])
    assert t == int(elen[5])
    assert float((enc[5, :, :t] - e1[0, :, :t]).abs().max()) < 1e-3
    # permutation equivariance over the batch dimension
    perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4])
    encp, elenp = eng.encode(*eng.frontend(wav[perm].contiguous(), wlen[perm]))
    assert torch.equal(elenp.cpu(), elen.cpu()[perm])
    vm = (torch.arange(501)[None, :] < elenp.cpu()[:, None])[:, None, :].to(enc.device)
    assert float(((encp - enc[perm]) * vm).abs().max()) < 1e-4


def test_fullsize_split_fp16_vs_exact_fp32_gemm(full_model):
    """The default split-fp16 MFMA path against the exact-fp32 MFMA path on the full
    16-layer model: same decode, activations within fp32 round-off of each other."""
    eng, wav, wlen = full_model
    assert eng.gemm_mode == "f16x3"
    feat, flen = eng.frontend(wav[:4], wlen[:4])
    enc, elen = eng.encode(feat, flen)
    dec = ragged_from_device(*eng.ctc_greedy(enc, elen))
    eng.set_gemm_mode("f32")
    try:
        feat32, _ = eng.frontend(wav[:4], wlen[:4])
        enc32, elen32 = eng.encode(feat32, flen)
        dec32 = ragged_from_device(*eng.ctc_greedy(enc32, elen32))
    finally:
        eng.set_gemm_mode("f16x3")
    vm = (torch.arange(enc.shape[2], device=enc.device)[None, :] < elen[:, None])[:, None, :]
    assert float(((feat - feat32)).abs().max()) < 2e-3
    assert float(((enc - enc32) * vm).abs().max()) < 2e-4
    assert dec == dec32


def test_fullsize_matches_oracle_on_one_utterance(full_model):
    """One 16-layer, 20 s utterance against the CPU oracle (a few seconds of CPU time)."""
    from gigaam_amd import synth
    from oracle import gigaam_oracle as O
    eng, wav, wlen = full_model
    ck = synth.make_checkpoint("v2_ctc", seed=0)
    with torch.no_grad():
        dec_o, enc_o, elen_o = O.transcribe_ids(ck, wav[:1], wlen[:1])
        lp = O.ctc_log_probs(ck["state_dict"], enc_o)
    enc, elen = eng.encode(*eng.frontend(wav[:1], wlen[:1]))
    assert float((enc.cpu() - enc_o).abs().max()) < 2e-3
    got = ragged_from_device(*eng.ctc_greedy(enc, elen))
    top2 = lp.topk(2, dim=-1).values
    margin = float((top2[..., 0] - top2[..., 1]).min())
    assert got == dec_o or margin < 1e-3, margin
