"""GPU: packed rows (gam_encode_varlen, gigaam_amd/csrc/gam_pack.h) -- a ragged batch whose lengths the caller also has on the host runs its
Conformer layers on the valid frames only.  The reference computes (and masks) every padded frame (gigaam/encoder.py:605-647; its optional
flash-attn path, gigaam/utils.py:103-155, unpads for attention alone), so the oracle here is the library's own padded path, which the golden /
live tests pin to the reference: the packed result must equal it on every valid frame -- bit for bit where the GEMM tiling has no split-K slices,
within 2e-5 where the slice count depends on the row count -- and the reference-anchored ragged tests (tests/test_hip_vs_reference_live.py,
test_hip_fullsize.py) run through the packed path too, because they hand their lengths over as CPU tensors."""
import pytest
import torch

from common import report

pytestmark = pytest.mark.gpu


def _engine(name, n_layers=2, **kw):
    from gigaam_amd import synth
    from gigaam_amd.engine import HipEngine, build_config
    import os
    ck = synth.make_checkpoint(name, seed=3, n_layers=n_layers, **kw)
    cfg = ck["cfg"]
    os.environ["GAM_PACK"] = "2"      # (read by gam_create) packed rows also in the small-batch / hipGraph regime, where the product keeps padded rows
    try:
        return HipEngine(build_config(cfg["preprocessor"], cfg["encoder"], cfg.get("head")), ck["state_dict"], torch.device("cuda:0")), ck
    finally:
        del os.environ["GAM_PACK"]


LENS_S = [3.0, 0.33, 2.01, 1.27, 2.56, 0.05, 2.999]     # incl. a 5-frame utterance and tile-edge lengths


@pytest.mark.parametrize("name", ["v2_ctc", "v3_e2e_ctc", "v1_ctc"])
@pytest.mark.parametrize("mode", ["f16x3", "f32", "f16"])
def test_packed_rows_equal_padded_rows(name, mode):
    from gigaam_amd import synth
    eng, _ = _engine(name)
    eng.set_gemm_mode(mode)
    lens = [int(16000 * s) for s in LENS_S]
    wav, wlen = synth.synth_audio(len(lens), max(LENS_S), seed=11, lengths=lens)
    feat, flen = eng.frontend(wav, wlen)
    host = flen.cpu().tolist()
    assert host == eng.host_feat_lengths(wlen)                      # the host formula is the device's
    for n_layers in (0, 1, -1):
        enc_p, elen_p, tok_p = eng.encode(feat, flen, n_layers_run=n_layers, want_tokens=True)                       # padded rows (no host lengths)
        assert eng.last_encode_rows()[0] == eng.last_encode_rows()[1]
        enc_k, elen_k, tok_k = eng.encode(feat, flen, n_layers_run=n_layers, want_tokens=True, host_lengths=host)    # packed rows
        rows, rows_pad = eng.last_encode_rows()
        assert rows == int(elen_k.clamp(max=enc_k.shape[2]).sum()) < rows_pad
        assert torch.equal(elen_p, elen_k)
        worst = 0.0
        for b, n in enumerate(elen_p.cpu().tolist()):
            d = float((enc_p[b, :, :n] - enc_k[b, :, :n]).abs().max()) if n else 0.0
            worst = max(worst, d, float((tok_p[b, :n] - tok_k[b, :n]).abs().max()) if n else 0.0)
            assert float(enc_k[b, :, n:].abs().max()) == 0.0 if n < enc_k.shape[2] else True     # zeros behind the last frame
            assert float(tok_k[b, n:].abs().max()) == 0.0 if n < tok_k.shape[1] else True
        report("packed_vs_padded", model=name, mode=mode, n_layers=n_layers, max_abs=worst)
        # (the opt-in fp16 speed mode rounds every GEMM operand to fp16: a last-bit difference of a split-K sum can move a rounding -- 5e-3)
        assert worst <= (5e-3 if mode == "f16" else 2e-5), (name, mode, n_layers, worst)
    # an upper bound is enough ...
    enc_u, _ = eng.encode(feat, flen, host_lengths=[h + 7 for h in host])
    n0 = int(elen_p[0])
    assert float((enc_u[0, :, :n0] - enc_p[0, :, :n0]).abs().max()) <= (5e-3 if mode == "f16" else 2e-5)
    torch.cuda.synchronize()
    assert eng.range_flag() is False


def test_packed_rows_bit_identical_without_splitk_and_faster_rows():
    """32 ragged utterances (linspace(4 s, 8 s)): large enough that no GEMM is split along K, so packing changes nothing but which rows exist.
    Built with the product's defaults (no GAM_PACK): 6432 token rows are above the hipGraph regime, so the batch IS packed."""
    from gigaam_amd import synth
    from gigaam_amd.engine import HipEngine, build_config
    ck = synth.make_checkpoint("v2_ctc", seed=3, n_layers=2)
    eng = HipEngine(build_config(ck["cfg"]["preprocessor"], ck["cfg"]["encoder"], ck["cfg"].get("head")), ck["state_dict"], torch.device("cuda:0"))
    lens = [int(16000 * (4.0 + 4.0 * i / 31)) for i in range(32)]
    wav, wlen = synth.synth_audio(32, 8.0, seed=5, lengths=lens)
    feat, flen = eng.frontend(wav, wlen)
    enc_p, elen = eng.encode(feat, flen)
    enc_k, _ = eng.encode(feat, flen, host_lengths=flen.cpu().tolist())
    rows, rows_pad = eng.last_encode_rows()
    assert rows == int(elen.sum()) and rows < 0.8 * rows_pad           # packed by default at this size
    same = all(torch.equal(enc_p[b, :, :n], enc_k[b, :, :n]) for b, n in enumerate(elen.cpu().tolist()))
    worst = max(float((enc_p[b, :, :n] - enc_k[b, :, :n]).abs().max()) for b, n in enumerate(elen.cpu().tolist()))
    report("packed_vs_padded_b32", bit_identical=bool(same), max_abs=worst)
    assert same, worst


def test_host_lengths_shorter_than_the_device_lengths_are_reported():
    from gigaam_amd import synth
    from gigaam_amd.engine import GigaAMHipError
    eng, _ = _engine("v2_ctc", n_layers=1)
    lens = [int(16000 * s) for s in (2.0, 1.0, 0.7, 1.5)]
    wav, wlen = synth.synth_audio(4, 2.0, seed=2, lengths=lens)
    feat, flen = eng.frontend(wav, wlen)
    host = flen.cpu().tolist()
    host[1] -= 40                                                # ten encoder frames short
    eng.encode(feat, flen, host_lengths=host)
    with pytest.raises(GigaAMHipError, match="host length"):
        eng.range_flag()
    assert eng.range_flag() is False                             # (cleared by the read)


def test_model_api_packs_when_lengths_arrive_on_the_cpu():
    """model.transcribe_batch / launch_batch with CPU lengths (what load_audio + collate produce) == the same call with GPU lengths (padded rows)."""
    import gigaam_amd
    from gigaam_amd import synth
    import os
    ck = synth.make_checkpoint("v2_rnnt", seed=1, n_layers=2)
    os.environ["GAM_PACK"] = "2"
    try:
        model = gigaam_amd.model_from_checkpoint(ck, "cuda:0")
    finally:
        del os.environ["GAM_PACK"]
    lens = [int(16000 * s) for s in (3.0, 1.1, 2.2, 0.4, 2.9, 1.7)]
    wav, wlen = synth.synth_audio(6, 3.0, seed=8, lengths=lens)
    eng = model.encoder.engine
    got_cpu = model.transcribe_batch(wav, wlen)
    r_cpu = eng.last_encode_rows()
    got_gpu = model.transcribe_batch(wav.cuda(), wlen.cuda())
    r_gpu = eng.last_encode_rows()
    assert got_cpu == got_gpu
    assert r_cpu[0] < r_cpu[1] and r_gpu[0] == r_gpu[1]      # CPU lengths: packed; GPU lengths: padded (no sync is made to fetch them)


def test_packed_rows_with_an_utterance_of_zero_frames():
    """v3 frontend (center = False): 160 samples are shorter than one window -> zero feature frames -> zero encoder frames.  The packed index
    gives that utterance no rows; every kernel must cope (cu[b] == cu[b + 1]) and the other utterances must be unaffected."""
    from gigaam_amd import synth
    eng, _ = _engine("v3_e2e_ctc")
    lens = [32000, 160, 16000, 100, 24000]
    wav, wlen = synth.synth_audio(len(lens), 2.0, seed=4, lengths=lens)
    feat, flen = eng.frontend(wav, wlen)
    host = flen.cpu().tolist()
    assert host[1] <= 0 and host[3] <= 0           # (the reference's out_len floors: 100 samples -> -1, gigaam/preprocess.py:86-92; clamped downstream)
    assert eng.host_feat_lengths(wlen)[3] == 0
    enc_p, elen = eng.encode(feat, flen)
    enc_k, elen_k = eng.encode(feat, flen, host_lengths=host)
    assert torch.equal(elen, elen_k) and elen.cpu().tolist()[1] == 0
    assert bool(torch.isfinite(enc_k).all())
    for b, n in enumerate(elen.cpu().tolist()):
        if n:
            assert float((enc_p[b, :, :n] - enc_k[b, :, :n]).abs().max()) <= 2e-5, b
    dec = eng.ctc_greedy(enc_k, elen_k)
    from gigaam_amd.engine import HipEngine
    rows, flag = HipEngine.collect(dec)
    assert rows[1] == ([], []) and rows[3] == ([], []) and not flag


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_packed_rows_on_random_batches(seed):
    """Seeded random batch sizes and lengths (a third of the utterances very short, some equal, the longest anywhere in the batch), rotary and
    rel-pos attention, BatchNorm and LayerNorm conv modules: packed == padded on every valid frame, CTC ids equal."""
    import random
    from gigaam_amd import synth
    from gigaam_amd.engine import HipEngine
    rng = random.Random(1234 + seed)
    name = ["v2_ctc", "v3_e2e_ctc", "v1_ctc", "v2_ctc"][seed]
    eng, _ = _engine(name)
    for _ in range(3):
        b = rng.choice([2, 3, 5, 9, 17, 33])
        tmax = rng.choice([1.0, 2.5, 4.0])
        lens = []
        for _ in range(b):
            r = rng.random()
            s = rng.uniform(0.03, 0.2) if r < 0.33 else (tmax if r > 0.9 else rng.uniform(0.2, tmax))
            lens.append(max(400, int(16000 * s)))
        if len(set(lens)) == 1:
            lens[0] = max(400, lens[0] // 2)
        wav, wlen = synth.synth_audio(b, max(lens) / 16000.0, seed=seed, lengths=lens)
        feat, flen = eng.frontend(wav, wlen)
        host = flen.cpu().tolist()
        enc_p, elen = eng.encode(feat, flen)
        ids_p, _ = HipEngine.collect(eng.ctc_greedy(enc_p, elen))
        enc_k, elen_k = eng.encode(feat, flen, host_lengths=host)
        rows, rows_pad = eng.last_encode_rows()
        ids_k, flag = HipEngine.collect(eng.ctc_greedy(enc_k, elen_k))
        assert torch.equal(elen, elen_k) and not flag
        n_valid = int(elen.clamp(min=0, max=enc_p.shape[2]).sum())
        assert rows == n_valid or rows == rows_pad        # (under 3 % padding the batch keeps padded rows)
        worst = max([float((enc_p[i, :, :n] - enc_k[i, :, :n]).abs().max()) for i, n in enumerate(elen.cpu().tolist()) if n > 0] or [0.0])
        report("packed_vs_padded_random", model=name, batch=b, rows=rows, rows_padded=rows_pad, max_abs=worst)
        assert worst <= 2e-5, (name, b, lens, worst)
        assert ids_p == ids_k
        assert bool(torch.isfinite(enc_k).all())


def test_batches_beyond_the_pack_index_capacity_keep_padded_rows():
    """The pack index is built by one workgroup with a 1024-entry prefix table: 1024 utterances pack, 1030 keep the padded layout -- same result."""
    from gigaam_amd import synth
    eng, _ = _engine("v2_ctc")
    for b in (1024, 1030):
        lens = [int(16000 * (0.25 + 0.35 * ((7 * i + 3) % 11) / 10)) for i in range(b)]
        wav, wlen = synth.synth_audio(b, max(lens) / 16000.0, seed=9, lengths=lens)
        feat, flen = eng.frontend(wav, wlen)
        enc_p, elen = eng.encode(feat, flen)
        enc_k, elen_k = eng.encode(feat, flen, host_lengths=flen.cpu().tolist())
        rows, rows_pad = eng.last_encode_rows()
        assert (rows < rows_pad) == (b <= 1024), (b, rows, rows_pad)
        assert torch.equal(elen, elen_k)
        n = elen.cpu().tolist()
        worst = max(float((enc_p[i, :, :n[i]] - enc_k[i, :, :n[i]]).abs().max()) for i in range(0, b, 37))
        assert worst <= 2e-5, (b, worst)
        assert bool(torch.isfinite(enc_k).all())
    assert eng.range_flag() is False
