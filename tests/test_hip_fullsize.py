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


def _fullsize_case(name):
    """(engine, ckpt, wav, wlen, golden) of a full-size golden (tests/golden/make_fullsize_golden.py: the REFERENCE's
    16-layer modules on utterances of the batches bench.py times; margins > 1e-3 (CTC) / 2e-3 (RNN-T) by selection)."""
    import json
    import os

    import numpy as np
    from common import ROOT
    from gigaam_amd import synth, workloads
    from gigaam_amd.engine import HipEngine, build_config
    gdir = os.path.join(ROOT, "tests", "golden")
    meta = json.load(open(os.path.join(gdir, "fullsize_meta.json")))[name]
    gold = dict(np.load(os.path.join(gdir, name + ".npz")))
    ck = synth.make_checkpoint(meta["model"], seed=0, rnnt_blank_bias=meta.get("blank_bias"))
    if name == "fullsize_v3_e2e_rnnt":
        wav, wlen = workloads.config4_batches(n_utts=1024, batch=32, only_batches=[0])[0][:2]
    else:
        wav, wlen = workloads.config2_batch(32, 20.0, rank=0)
    keep = gold["utt_index"].tolist()
    wav, wlen = wav[keep].contiguous(), wlen[keep].contiguous()
    assert wlen.tolist() == gold["wav_len"].tolist()
    cfg = ck["cfg"]
    eng = HipEngine(build_config(cfg["preprocessor"], cfg["encoder"], cfg["head"]), ck["state_dict"], torch.device("cuda:0"))
    return eng, ck, wav, wlen, gold, meta


def _ref_ragged(gold):
    from common import split_ragged
    return split_ragged(gold["ids"], gold["frames"], gold["counts"].tolist())


def test_fullsize_config2_ctc_matches_reference():
    """BASELINE config 2 at full size: 4 utterances of the timed batch, 16 layers, 20 s -- encoder output against the
    reference's (probe), ids + frames bit-exact.  No escape hatch: the utterances were chosen with margins > 1e-3."""
    from common import report
    eng, ck, wav, wlen, gold, meta = _fullsize_case("fullsize_v2_ctc")
    enc, elen = eng.encode(*eng.frontend(wav, wlen))
    assert elen.cpu().tolist() == gold["enc_len"].tolist()
    err = float((enc.cpu()[:, ::16, ::5] - torch.from_numpy(gold["enc_probe"])).abs().max())
    report("fullsize_encoder_vs_reference", case="fullsize_v2_ctc", err=err, tol=2e-4, min_margin=meta["min_margin"])
    assert err < 2e-4, err   # measured 8e-6 (r02)
    assert ragged_from_device(*eng.ctc_greedy(enc, elen)) == _ref_ragged(gold)


@pytest.mark.parametrize("name", ["fullsize_v2_rnnt", "fullsize_v3_e2e_rnnt"])
def test_fullsize_rnnt_matches_reference(name):
    """BASELINE configs 3 / 4 at full size (16 layers, blank-dominant head, V = 34 / 1025): ids + frames exact, the number
    of joint evaluations exact, and the top-4 log-probs of EVERY joint step within 1e-3 of the reference's."""
    from common import report
    eng, ck, wav, wlen, gold, meta = _fullsize_case(name)
    ms = ck["cfg"]["decoding"]["max_symbols_per_step"]
    enc, elen = eng.encode(*eng.frontend(wav, wlen))
    assert elen.cpu().tolist() == gold["enc_len"].tolist()
    err = float((enc.cpu()[:, ::16, ::5] - torch.from_numpy(gold["enc_probe"])).abs().max())
    report("fullsize_encoder_vs_reference", case=name, err=err, tol=2e-4, min_margin=meta["min_margin"])
    assert err < 2e-4, err   # measured 8e-6 (r02)
    tc = gold["trace_counts"].tolist()
    ids, frames, counts, dump, dcount = eng.rnnt_greedy(enc, elen, ms, dump_cap=max(tc))
    assert ragged_from_device(ids, frames, counts) == _ref_ragged(gold)
    assert dcount.cpu().tolist() == tc
    o = 0
    worst = 0.0
    for i, c in enumerate(tc):
        want_v = torch.from_numpy(gold["trace_top_vals"][o:o + c])
        want_i = torch.from_numpy(gold["trace_top_idx"][o:o + c]).long()
        o += c
        got_v = dump[i, :c].cpu().gather(1, want_i)
        worst = max(worst, float((got_v - want_v).abs().max()))
    report("fullsize_rnnt_joint_logprobs", case=name, err=worst, tol=1e-3, steps=sum(tc), symbols_per_frame=meta["symbols_per_frame"])
    assert worst < 1e-3, worst


def test_fullsize_config5_longform_chunks_match_reference():
    """BASELINE config 5 at full size: the REFERENCE's 16-layer modules on three chunks decoded as one
    zero-padded batch, as transcribe_longform would -- a 30 s window of the hour (the packer's strict limit: T' = 751, the
    longest chunk the path can ever see), the longest chunk of the list bench.py --config 5 times and a second long one
    (tests/golden/make_longform_golden.py; margins > 1e-3 by selection).  Encoder probe <= 2e-4, ids + frames bit-exact."""
    import json
    import os

    import numpy as np
    from common import ROOT, report
    from gigaam_amd import synth, workloads
    from gigaam_amd.engine import HipEngine, build_config
    from gigaam_amd.feeder import collate
    gdir = os.path.join(ROOT, "tests", "golden")
    meta = json.load(open(os.path.join(gdir, "fullsize_meta.json")))["fullsize_v2_ctc_longform"]
    gold = dict(np.load(os.path.join(gdir, "fullsize_v2_ctc_longform.npz")))
    segs, _ = workloads.config5_segments(3600)
    assert len(segs) == meta["n_chunks"]
    audio = workloads.config5_audio(3600)
    o0 = int(meta["strict_window_offset_s"] * 16000)
    chunks = [audio[o0: o0 + int(meta["strict_window_s"] * 16000)]] + [segs[i] for i in meta["chunk_index"][1:]]
    wav, wlen = collate(chunks)
    assert wlen.tolist() == gold["wav_len"].tolist()
    ck = synth.make_checkpoint("v2_ctc", seed=0)
    cfg = ck["cfg"]
    eng = HipEngine(build_config(cfg["preprocessor"], cfg["encoder"], cfg["head"]), ck["state_dict"], torch.device("cuda:0"))
    enc, elen = eng.encode(*eng.frontend(wav, wlen))
    assert elen.cpu().tolist() == gold["enc_len"].tolist() == [751, 550, 546]
    vm = (torch.arange(enc.shape[2])[None, :] < elen.cpu()[:, None])[:, None, ::5]      # ragged batch: valid frames only
    err = float(((enc.cpu()[:, ::16, ::5] - torch.from_numpy(gold["enc_probe"])) * vm).abs().max())
    report("fullsize_encoder_vs_reference", case="fullsize_v2_ctc_longform", err=err, tol=2e-4, min_margin=meta["min_margin"])
    assert err < 2e-4, err
    assert ragged_from_device(*eng.ctc_greedy(enc, elen)) == _ref_ragged(gold)


def test_fullsize_config2_whole_batch_matches_reference():
    """The WHOLE timed batch of BASELINE config 2: all 32 utterances of 20 s through the REFERENCE's 16-layer
    modules as one batch of 32 (tests/golden/make_fullsize32_golden.py).  Encoder probe of every utterance <= 2e-4; ids + frames
    bit-exact for ALL 32 utterances (r04 allowed an edit distance of 2 on the 7 utterances whose smallest reference top-1 / top-2
    margin is below 5e-4 and measured 32 / 32; the escape hatch is gone -- the margins still go to the report)."""
    import json
    import os

    import numpy as np
    from common import ROOT, report, split_ragged
    from gigaam_amd import synth, workloads
    from gigaam_amd.engine import HipEngine, build_config
    gdir = os.path.join(ROOT, "tests", "golden")
    path = os.path.join(gdir, "fullsize32_v2_ctc.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/fullsize32_v2_ctc.npz not generated")
    gold = dict(np.load(path))
    ck = synth.make_checkpoint("v2_ctc", seed=0)
    cfg = ck["cfg"]
    eng = HipEngine(build_config(cfg["preprocessor"], cfg["encoder"], cfg["head"]), ck["state_dict"], torch.device("cuda:0"))
    wav, wlen = workloads.config2_batch(32, 20.0, rank=0)
    enc, elen = eng.encode(*eng.frontend(wav, wlen))
    assert elen.cpu().tolist() == gold["enc_len"].tolist()
    err = float((enc.cpu()[:, ::16, ::5] - torch.from_numpy(gold["enc_probe"])).abs().max())
    got = ragged_from_device(*eng.ctc_greedy(enc, elen))
    ref = split_ragged(gold["ids"], gold["frames"], gold["counts"].tolist())
    margins = gold["min_margin"].tolist()
    same = [g == r for g, r in zip(got, ref)]
    report("fullsize32_vs_reference", err=err, tol=2e-4, utterances_identical=f"{sum(same)}/32",
           margins_of_differing=[round(m, 6) for m, s in zip(margins, same) if not s], min_margin=min(margins))
    assert err < 2e-4, err
    # measured since r04: 32 / 32 bit-exact, near-tie utterances included -- assert what is measured; a
    # future regression on a 2e-5-margin frame fails loudly here and is then judged with its margin in hand
    for i, (g, r, m) in enumerate(zip(got, ref, margins)):
        assert g == r, (i, "reference min margin", m)
    assert sum(1 for m in margins if m > 5e-4) >= 24      # the fixture is not vacuous
