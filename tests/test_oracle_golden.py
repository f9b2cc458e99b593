"""CPU: the oracle restatement reproduces the REFERENCE's committed outputs
(tests/golden/*.npz were produced by the reference's own modules, see make_golden.py)."""
import numpy as np
import pytest
import torch

from common import CASES, RNNT_MIN_MARGIN, golden_trace, load_case, oracle_features, split_ragged, valid_mask
from oracle import gigaam_oracle as O


@pytest.mark.parametrize("case", list(CASES))
def test_oracle_matches_reference_golden(case):
    torch.set_num_threads(8)
    ck, wav, wlen, gold = load_case(case)
    cfg, sd = ck["cfg"], ck["state_dict"]
    feat, flen = oracle_features(ck, wav, wlen)
    assert flen.tolist() == gold["feat_len"].tolist()
    np.testing.assert_allclose(feat[:, ::7, ::13].numpy(), gold["feat_probe"], atol=1e-5)
    stages = {}
    with torch.no_grad():
        enc, elen = O.encoder_forward(sd, cfg["encoder"], feat, flen, stages=stages)
    assert elen.dtype == torch.int32 and elen.tolist() == gold["enc_len"].tolist()
    vm = valid_mask(enc.shape[2], elen)
    assert float(((stages["pre_encode"] - torch.from_numpy(gold["pre_encode"])) * vm[:, :, None]).abs().max()) < 2e-5
    assert float(((enc - torch.from_numpy(gold["encoded"])) * vm[:, None, :]).abs().max()) < 2e-5
    ref = split_ragged(gold["ids"], gold["frames"], gold["counts"].tolist())
    enc_ref = torch.from_numpy(gold["encoded"])
    with torch.no_grad():
        if "log_probs" in gold:
            lp = O.ctc_log_probs(sd, enc_ref)
            assert float((lp - torch.from_numpy(gold["log_probs"])).abs().max()) < 2e-5
            got = O.ctc_greedy(lp, elen)
        else:
            trace = []
            got = O.rnnt_greedy(sd, enc_ref, elen, cfg["decoding"]["max_symbols_per_step"],
                                cfg["head"]["decoder"]["pred_rnn_layers"], trace=trace)
            # every joint evaluation against the reference's own (RNNTJoint.joint recorded during its decode)
            for i, want in enumerate(golden_trace(gold)):
                mine = torch.stack([t[2] for t in trace if t[0] == i])
                assert mine.shape == want.shape and float((mine - want).abs().max()) < 2e-5
                top2 = want.topk(2, dim=-1).values
                assert float((top2[:, 0] - top2[:, 1]).min()) > RNNT_MIN_MARGIN   # no near-tie anywhere in the fixture
    assert got == ref


@pytest.mark.parametrize("case", list(CASES))
def test_frontend_against_torchaudio_fixture(case):
    """Row a1 against torchaudio's own MelSpectrogram + the reference's SpecScaler, when the fixture exists."""
    from common import frontend_fixture
    fx = frontend_fixture(case)
    if fx is None:
        pytest.skip("a1 parity unpinned: no torchaudio here; run tests/golden/make_frontend_golden.py where it is installed")
    ck, wav, wlen, _ = load_case(case)
    feat, flen = oracle_features(ck, wav, wlen)
    assert flen.tolist() == fx["feat_len"].tolist() and tuple(feat.shape) == fx["feat"].shape
    from common import TOL_FEAT, TOL_FEAT_WEAK, logmel_err
    fm = valid_mask(feat.shape[2], flen)
    e_strong, e_weak = logmel_err(feat, torch.from_numpy(fx["feat"]), fm)
    assert e_strong < TOL_FEAT and e_weak < TOL_FEAT_WEAK, (e_strong, e_weak)


@pytest.mark.parametrize("model", ["v2_ctc", "v3_ctc"])
def test_frontend_against_transformers_audio_utils(model):
    """A third independent statement of the same published algorithm: Hugging Face's ``transformers.audio_utils``
    (periodic Hann, reflect padding / center=False, power spectrogram, HTK filterbank with norm=None, log with a floor) --
    the implementation HF's feature extractors use, evaluated in fp64.  Not the reference either (row a1 stays
    "parity unpinned" until tests/golden/make_frontend_golden.py runs next to torchaudio), but the restatement, scipy.fft
    and this library agree on frame counts, the filterbank (3e-8) and the log-mel values (<= 1e-4 / 4e-4)."""
    audio_utils = pytest.importorskip("transformers.audio_utils")
    from gigaam_amd import synth
    cfg = synth.model_cfg(model)["preprocessor"]
    n_fft, win, hop = cfg.get("n_fft", 400), cfg.get("win_length", 400), cfg.get("hop_length", 160)
    center = cfg.get("center", True)
    wav, wlen = synth.synth_audio(1, 2.0, seed=3)
    fb = audio_utils.mel_filter_bank(num_frequency_bins=n_fft // 2 + 1, num_mel_filters=64, min_frequency=0.0, max_frequency=8000.0,
                                     sampling_rate=16000, norm=None, mel_scale="htk")
    spec = audio_utils.spectrogram(wav[0].numpy().astype(np.float64), audio_utils.window_function(win, "hann", periodic=True),
                                   frame_length=win, hop_length=hop, fft_length=n_fft, power=2.0, center=center, pad_mode="reflect",
                                   onesided=True, mel_filters=fb, mel_floor=1e-9, log_mel="log")
    win_t = torch.from_numpy(synth.hann_window_periodic(win))
    fb_t = torch.from_numpy(synth.mel_filterbank_htk(n_fft // 2 + 1, 64, 16000))
    assert float(np.abs(fb - fb_t.numpy()).max()) < 1e-6
    feat, flen = O.log_mel(wav, wlen, cfg, win_t, fb_t)
    assert spec.shape == tuple(feat.shape[1:]) and int(flen[0]) == spec.shape[1]
    f = feat[0].numpy().astype(np.float64)
    strong = f >= f.max(axis=0, keepdims=True) - 60.0 * 0.2302585
    d = np.abs(spec - f)
    assert d[strong].max() < 3e-4 and d.max() < 2e-3, (d[strong].max(), d.max())


def test_emotion_head_oracle_matches_reference_golden():
    """GigaAMEmo (model.py:272-293): tests/golden/emo_l2.npz holds what the reference's own get_probs /
    forward_for_export bodies returned (make_golden.py runs them unbound on the reference encoder)."""
    import os
    from common import EMO_CASE, ROOT
    from gigaam_amd import synth
    from common import make_case_checkpoint
    ck, wav, wlen = make_case_checkpoint(EMO_CASE)
    b = wav.shape[0]
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "emo_l2.npz")))
    sd, cfg = ck["state_dict"], ck["cfg"]
    torch.set_num_threads(8)
    with torch.no_grad():
        feat, flen = oracle_features(ck, wav, wlen)
        enc, elen = O.encoder_forward(sd, cfg["encoder"], feat, flen)
        assert float((O.emo_probs(sd, enc) - torch.from_numpy(gold["probs_export"])).abs().max()) < 1e-5
        for i in range(b):   # one unpadded file at a time, as get_probs does
            f1, l1 = oracle_features(ck, wav[i:i + 1, : int(wlen[i])], wlen[i:i + 1])
            e1, _ = O.encoder_forward(sd, cfg["encoder"], f1, l1)
            p = O.emo_probs(sd, e1)[0]
            assert abs(float(p.sum()) - 1.0) < 1e-6
            assert float((p - torch.from_numpy(gold["probs_single"][i])).abs().max()) < 1e-5


def test_frontend_known_answers():
    """a1 is parity-unpinned (no torchaudio here): analytic checks of the restatement."""
    from gigaam_amd import synth
    cfg = synth.model_cfg("v2_ctc")["preprocessor"]
    win = torch.from_numpy(synth.hann_window_periodic(400))
    fb = torch.from_numpy(synth.mel_filterbank_htk(201, 64, 16000))
    # frame-count formula (reference preprocess.py:78-92)
    for n in (3200, 16000, 80000, 80001, 320000):
        wav = torch.zeros(1, n)
        feat, flen = O.log_mel(wav, torch.tensor([n]), cfg, win, fb)
        assert feat.shape == (1, 64, n // 160 + 1) and int(flen) == n // 160 + 1
        assert torch.allclose(feat, torch.full_like(feat, float(np.log(1e-9))))  # silence -> clamp floor
    # pure tone at an exact bin (k=40 -> 1600 Hz): energy sits in the mel bands around it
    t = torch.arange(16000) / 16000.0
    feat, _ = O.log_mel(torch.sin(2 * np.pi * 1600.0 * t)[None], torch.tensor([16000]), cfg, win, fb)
    mid = feat[0, :, 50]
    band = int(mid.argmax())
    assert fb[40, band] > 0 and fb[40].argmax() == band
    # Hann-windowed bin-centred tone: |X_k|^2 = (N/4)^2 and (N/8)^2 in the two neighbours
    expect = 100.0 ** 2 * float(fb[40, band]) + 50.0 ** 2 * float(fb[39, band] + fb[41, band])
    assert abs(float(mid.max()) - float(np.log(expect))) < 1e-3
    # filterbank: triangles cover every non-edge bin, all weights in [0, 1]
    assert float(fb.min()) >= 0 and float(fb.max()) <= 1.0
    assert bool((fb[1:-1].sum(dim=1) > 0).all())
    # v3 frontend: center=False, 320/160
    cfg3 = synth.model_cfg("v3_ctc")["preprocessor"]
    feat, flen = O.log_mel(torch.zeros(1, 16000), torch.tensor([16000]), cfg3,
                           torch.from_numpy(synth.hann_window_periodic(320)),
                           torch.from_numpy(synth.mel_filterbank_htk(161, 64, 16000)))
    assert feat.shape[2] == (16000 - 320) // 160 + 1 == int(flen)


def test_frontend_against_independent_scipy_stft():
    """a1 has no reference-produced golden (torchaudio is absent here): besides the analytic checks above,
    the restatement is compared with an INDEPENDENT fp64 implementation built on scipy.signal (framing by
    hand, reflect padding, periodic Hann, rfft) -- a different code path from torch.stft."""
    import scipy.fft
    from gigaam_amd import synth
    for name, n_fft, center in (("v2_ctc", 400, True), ("v3_ctc", 320, False)):
        cfg = synth.model_cfg(name)["preprocessor"]
        win = synth.hann_window_periodic(n_fft)
        fbank = synth.mel_filterbank_htk(n_fft // 2 + 1, 64, 16000)
        wav, lens = synth.synth_audio(2, 1.5, seed=77, lengths=[24000, 17000])
        feat, flen = O.log_mel(wav, lens, cfg, torch.from_numpy(win), torch.from_numpy(fbank))
        x = wav.numpy().astype(np.float64)
        if center:
            x = np.pad(x, ((0, 0), (n_fft // 2, n_fft // 2)), mode="reflect")
        n_frames = (x.shape[1] - n_fft) // 160 + 1
        frames = np.stack([x[:, i * 160: i * 160 + n_fft] for i in range(n_frames)], axis=1) * win.astype(np.float64)
        power = np.abs(scipy.fft.rfft(frames, axis=-1)) ** 2                       # [B, T, n_freq]
        ref = np.log(np.clip(power @ fbank.astype(np.float64), 1e-9, 1e9)).transpose(0, 2, 1)
        assert feat.shape == ref.shape and flen.tolist() == [(l // 160 + 1) if center else ((l - n_fft) // 160 + 1) for l in lens.tolist()]
        ref_t = torch.from_numpy(ref)
        strong = ref_t >= ref_t.max(dim=1, keepdim=True).values - 60.0 * 0.2302585   # see tests/common.py
        d = (feat.double() - ref_t).abs()
        assert float((d * strong).max()) < 5e-4 and float((d * ~strong).max()) < 2e-2


# ----------------------------------------------------------------------------- oracle/_ref: the reference itself, travelling
def _run_py(code, **env):
    import os
    import subprocess
    import sys
    from common import ROOT
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_reference_bytecode_reproduces_the_committed_goldens():
    """oracle/_ref (CPython bytecode of /root/reference/gigaam/*.py, oracle/build_ref.py) imported WITHOUT the source tree
    (GIGAAM_REF_FORCE_BYTECODE=1: what the GPU box sees) must BE the reference: its GigaAMASR.forward + decoding.decode on the
    cases' seeded checkpoints reproduces the committed reference outputs (encoder <= 2e-5: the only difference is the stand-in
    MelSpectrogram's |X|.abs().pow(2) vs the oracle's re^2 + im^2 in the features; ids + frames exact)."""
    import os
    from common import ROOT
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "BUILT.json")):
        pytest.skip("oracle/_ref not built (python oracle/build_ref.py needs /root/reference)")
    out = _run_py(r"""
import json, sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from common import load_case, split_ragged, valid_mask
from oracle import ref_shim
assert ref_shim.reference_kind() == "bytecode", ref_shim.reference_kind()
res = {}
torch.set_num_threads(8)
for case in ("v2_ctc_l2", "v1_ctc_l2", "v3_e2e_rnnt_l2", "v2_rnnt_l2_lstm2"):
    ck, wav, wlen, gold = load_case(case)
    m = ref_shim.reference_model(ck)
    assert type(m).__module__ == "gigaam.model" and m.encoder.__class__.__module__ == "gigaam.encoder"
    with torch.inference_mode():
        enc, elen = m.forward(wav, wlen)
        dec = m.decoding.decode(m.head, enc, elen)
    vm = valid_mask(enc.shape[2], elen)[:, None, :]
    err = float(((enc - torch.from_numpy(gold["encoded"])) * vm).abs().max())
    want = split_ragged(gold["ids"], gold["frames"], gold["counts"].tolist())
    res[case] = [err, [(list(i), list(f)) for _, i, f in dec] == want]
import gigaam
res["origin"] = gigaam.__file__
print(json.dumps(res))
""", GIGAAM_REF_FORCE_BYTECODE="1")
    import json
    res = json.loads(out.strip().splitlines()[-1])
    assert res.pop("origin").endswith(os.path.join("oracle", "_ref", "gigaam", "__init__.pyc"))
    for case, (err, same) in res.items():
        assert err < 2e-5 and same, (case, err, same)


def test_reference_manifest_matches_the_tree_it_was_built_from():
    """oracle/ref_manifest.json (committed) = sha256 of every reference source oracle/_ref was compiled from; where the
    reference tree is present (the build container) the files themselves must still hash to it."""
    import hashlib
    import json
    import os
    from common import ROOT
    man = json.load(open(os.path.join(ROOT, "oracle", "ref_manifest.json")))
    assert {"gigaam/encoder.py", "gigaam/decoder.py", "gigaam/decoding.py", "gigaam/model.py", "gigaam/preprocess.py",
            "gigaam/utils.py", "gigaam/onnx_utils.py"} <= set(man)
    built = os.path.join(ROOT, "oracle", "_ref", "BUILT.json")
    if os.path.exists(built):
        assert json.load(open(built))["sources_sha256"] == man
    from oracle import ref_shim
    if os.path.isdir(os.path.join(ref_shim.REFERENCE_ROOT, "gigaam")):
        for rel, h in man.items():
            assert hashlib.sha256(open(os.path.join(ref_shim.REFERENCE_ROOT, rel), "rb").read()).hexdigest() == h, rel


def test_onnx_twin_ctc_decoder_agrees_with_the_oracle():
    """gigaam/onnx_utils.py:39-54 (the reference's numpy statement of a11) on the golden log-probs == the golden ids."""
    from oracle import ref_shim
    if not ref_shim.reference_available():
        pytest.skip("no reference (oracle/_ref not built)")
    twins = ref_shim.import_onnx_twins()
    ref = ref_shim.import_reference()
    for case in ("v2_ctc_l2", "v3_e2e_ctc_l2"):
        ck, wav, wlen, gold = load_case(case)
        tok = ref.decoding.Tokenizer(ck["cfg"]["decoding"]["vocabulary"])
        texts = twins._decode_ctc_batch(gold["log_probs"].argmax(-1), gold["enc_len"], tok)
        want = split_ragged(gold["ids"], gold["frames"], gold["counts"].tolist())
        assert texts == [tok.decode(i) for i, _ in want]
