"""CPU: the C-ABI library loads and exports every declared symbol; host-side logic."""
import os
import re
import sys
import wave

import numpy as np
import pytest
import torch

from common import ROOT

import gigaam_amd
from gigaam_amd import _lib, synth


def test_library_exports_every_header_symbol():
    header = open(os.path.join(ROOT, "include", "gigaam_hip.h")).read()
    declared = set(re.findall(r"\b(gam_[a-z_0-9]+)\s*\(", header))
    declared -= {"gam_handle"}
    assert declared, "no declarations parsed"
    lib = _lib.load_library()
    assert lib.gam_abi_version() == 1
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))


def test_device_code_has_no_unreliable_packed_fp32():
    """r06: a v_pk_{fma,mul,add}_f32 whose low result reads the HIGH register of src1 (op_sel:[x,1,..]) miscomputes lanes 48..63 on MI355X
    whenever another wave on its SIMD issues MFMAs (tools/pkfma_rule.hip, profiles/r06_pkfma_rule.txt) -- the mechanism behind r05's
    "co-residency perturbation" of the RNN-T decode.  The shipped library must not contain one (gigaam_amd/build.py gates the build too),
    and the gate itself must recognise the form."""
    from gigaam_amd import build
    assert "-fno-slp-vectorize" in build.FLAGS
    assert build._RISKY.search("\tv_pk_fma_f32 v[4:5], v[134:135], v[174:175], v[4:5] op_sel:[0,1,0]")
    assert build._RISKY.search("\tv_pk_mul_f32 v[4:5], v[134:135], v[174:175] op_sel:[0,1]")
    assert not build._RISKY.search("\tv_pk_fma_f32 v[4:5], v[134:135], v[174:175], v[4:5] op_sel_hi:[1,0,1]")
    assert not build._RISKY.search("\tv_pk_fma_f32 v[4:5], v[134:135], v[174:175], v[4:5] op_sel:[1,0,1]")
    assert not build._RISKY.search("\tv_pk_fma_f16 v4, v134, v174, v4 op_sel:[0,1,0]")
    hits = build.risky_packed_f32(_lib.LIB_PATH)
    assert hits == [], hits[:10]


def test_config_mirror_matches_reference_defaults():
    from gigaam_amd.engine import build_config
    cfg = synth.model_cfg("v2_ctc")
    c = build_config(cfg["preprocessor"], cfg["encoder"], cfg["head"])
    assert (c.n_fft, c.win_length, c.hop_length, c.center) == (400, 400, 160, 1)
    assert (c.d_model, c.n_heads, c.n_layers, c.conv_kernel_size) == (768, 16, 16, 31)
    assert (c.head_type, c.num_classes) == (_lib.HEAD_CTC, 34)
    cfg = synth.model_cfg("v3_e2e_rnnt")
    c = build_config(cfg["preprocessor"], cfg["encoder"], cfg["head"])
    assert (c.n_fft, c.center, c.subsampling, c.conv_norm_type, c.conv_kernel_size) == (320, 0, _lib.SUBS_CONV1D, _lib.NORM_LAYER, 5)
    assert (c.head_type, c.num_classes, c.pred_hidden, c.joint_hidden) == (_lib.HEAD_RNNT, 1025, 320, 320)


def test_no_cpu_fallback():
    """Without a GPU the product path must fail loudly, never compute on the CPU."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ck = synth.make_checkpoint("v2_ctc", seed=1, n_layers=1)
    model = gigaam_amd.model_from_checkpoint(ck, "cpu")
    wav, wlen = synth.synth_audio(1, 0.5, seed=0)
    with pytest.raises(_lib.GigaAMHipError):
        model.forward(wav, wlen)


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gigaam_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                src = open(os.path.join(dirpath, f), encoding="utf-8").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "ref_shim" not in src and "build_ref" not in src and "oracle/_ref" not in src and "_ref" + os.sep not in src, f
    # ... nor ships any reference bytecode of its own: oracle/_ref is the only place the reference's modules live
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gigaam_amd")):
        assert "_ref" not in dirpath.split(os.sep)


def test_load_model_errors_and_api(tmp_path):
    with pytest.raises(ValueError):
        gigaam_amd.load_model("no_such_model", download_root=str(tmp_path))
    with pytest.raises(FileNotFoundError):
        gigaam_amd.load_model("v2_ctc", download_root=str(tmp_path))
    assert gigaam_amd.format_time(3725.5) == "01:02:05:50"
    assert gigaam_amd.format_time(65.25) == "01:05:25"


def test_load_audio_wav_and_errors(tmp_path):
    from gigaam_amd.preprocess import load_audio
    pcm = (np.sin(np.arange(1600) / 10.0) * 12000).astype(np.int16)
    p = str(tmp_path / "a.wav")
    with wave.open(p, "wb") as wf:
        wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(16000); wf.writeframes(pcm.tobytes())
    x = load_audio(p)
    assert x.dtype == torch.float32 and torch.equal(x, torch.from_numpy(pcm.astype(np.float32)) / 32768.0)
    with pytest.raises(RuntimeError, match="Failed to load audio"):
        load_audio(str(tmp_path / "missing.wav"))


def test_too_long_guard_and_result_types(tmp_path):
    from gigaam_amd.types import LongformTranscriptionResult, Segment, TranscriptionResult, Word
    r = TranscriptionResult(text="привет", words=[Word("привет", 0.0, 0.4)])
    assert str(r) == "привет"
    lf = LongformTranscriptionResult([Segment("а", 0, 1, [Word("а", 0, 1)]), Segment("б", 1, 2, [Word("б", 1, 2)])])
    assert lf.text == "а б" and len(lf) == 2 and lf.has_word_timestamps and [w.text for w in lf.words] == ["а", "б"]
    ck = synth.make_checkpoint("v2_ctc", seed=1, n_layers=1)
    model = gigaam_amd.model_from_checkpoint(ck, "cpu")
    p = str(tmp_path / "long.wav")
    with wave.open(p, "wb") as wf:
        wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(16000); wf.writeframes(np.zeros(16000 * 26, np.int16).tobytes())
    with pytest.raises(ValueError, match="Too long wav file"):
        model.transcribe(p)


def test_frames_to_words_and_vad_packing():
    from gigaam_amd.decoding import Tokenizer
    from gigaam_amd.timestamps_utils import compute_frame_shift, frames_to_words
    from gigaam_amd.vad_utils import pack_regions
    tok = Tokenizer(synth.CHAR_VOCAB)
    ids = [synth.CHAR_VOCAB.index(c) for c in "да нет"]
    frames = [3, 4, 7, 10, 11, 12]
    words = frames_to_words(tok, ids, frames, 0.04)
    assert [(w.text, round(w.start, 2), round(w.end, 2)) for w in words] == [("да", 0.12, 0.2), ("нет", 0.4, 0.52)]
    assert compute_frame_shift(320000, 500) == pytest.approx(0.04)
    sp = Tokenizer(["▁a", "b", "▁c"])
    assert [w.text for w in frames_to_words(sp, [0, 1, 2], [0, 1, 5], 0.04)] == ["ab", "c"]
    # packing: merge until > min_duration, split the 70 s region into 3 equal parts, drop a 0.1 s tail
    regs = [(0.0, 5.0), (5.5, 12.0), (12.5, 18.0), (18.4, 21.0), (30.0, 100.0), (100.5, 100.6)]
    out = pack_regions(regs, audio_seconds=101.0)
    assert out[0] == (0.0, 18.0)
    assert out[1] == (18.4, 21.0)
    assert len(out) == 5 and out[2][0] == 30.0 and abs(out[4][1] - 100.0) < 1e-9
    assert all(e - s <= 30.0 + 1e-9 for s, e in out)


def test_bench_power_sampler_without_telemetry(monkeypatch):
    """bench.py's board-power leg must degrade to a labelled "unavailable" record (never an exception) where neither
    the amdgpu hwmon files nor rocm-smi exist -- e.g. this CPU container."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench, "_hwmon_dir", lambda i: None)
    monkeypatch.setattr("shutil.which", lambda name: None)
    monkeypatch.setattr("os.path.exists", lambda p: False)
    with bench.PowerSampler(0) as ps:
        pass
    out = ps.summary()
    assert out["available"] is False and "note" in out


# --------------------------------------------------------------------------- real-checkpoint plumbing (reference __init__.py:139-192)
import collections.abc as _abc


class FakeDictConfig(_abc.MutableMapping):
    """Stand-in for omegaconf.DictConfig (absent offline): a MutableMapping with attribute access, nested containers
    wrapped on the way in -- what ``checkpoint["cfg"]`` is in the reference's .ckpt files."""

    def __init__(self, d):
        object.__setattr__(self, "_d", {k: _wrap(v) for k, v in d.items()})

    def __getitem__(self, k):
        return self._d[k]

    def __setitem__(self, k, v):
        self._d[k] = _wrap(v)

    def __delitem__(self, k):
        del self._d[k]

    def __iter__(self):
        return iter(self._d)

    def __len__(self):
        return len(self._d)

    def __getattr__(self, k):
        try:
            return object.__getattribute__(self, "_d")[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self._d[k] = _wrap(v)


class FakeListConfig(_abc.Sequence):
    def __init__(self, items):
        self._l = [_wrap(v) for v in items]

    def __getitem__(self, i):
        return self._l[i]

    def __len__(self):
        return len(self._l)


def _wrap(v):
    if isinstance(v, dict):
        return FakeDictConfig(v)
    if isinstance(v, (list, tuple)):
        return FakeListConfig(v)
    return v


def _real_style_checkpoint(name="v2_ctc"):
    """A synthetic checkpoint dressed like a published one: attribute-style cfg, BatchNorm's int64 ``num_batches_tracked``
    and the torchaudio buffers the reference's FeatureExtractor registers (SURVEY 8b state_dict keys)."""
    import torch
    from gigaam_amd import synth
    ck = synth.make_checkpoint(name, seed=3, n_layers=2)
    sd = dict(ck["state_dict"])
    for i in range(2):
        sd[f"encoder.layers.{i}.conv.batch_norm.num_batches_tracked"] = torch.tensor(1234, dtype=torch.int64)
    pre = ck["cfg"]["preprocessor"]
    n_fft = pre.get("n_fft", 400)
    sd["preprocessor.featurizer.0.spectrogram.window"] = torch.hann_window(n_fft, periodic=True)
    return ck, {"cfg": FakeDictConfig(ck["cfg"]), "state_dict": sd}


def test_attribute_style_cfg_plumbing(tmp_path, monkeypatch):
    """No real .ckpt has ever gone through load_model offline; the first user's will.  Everything load_model does to one is
    exercised here on a look-alike: cfg nodes by attribute AND by key, ListConfig vocabulary, ``cfg.decoding.model_path`` /
    ``cfg.model_name`` assignment (reference __init__.py:176,191), the md5 gate, and the Lightning fine-tuned form."""
    import hashlib
    import torch
    import gigaam_amd
    from gigaam_amd import model as M
    plain, ck = _real_style_checkpoint()
    cfg = ck["cfg"]
    assert cfg.encoder.d_model == cfg["encoder"]["d_model"] and not isinstance(cfg, dict)
    m = gigaam_amd.model_from_checkpoint(ck, "cpu")
    assert isinstance(m, gigaam_amd.GigaAMASR)
    assert m.encoder.cfg["n_layers"] == 2 and m.encoder.cfg["self_attention_model"] == plain["cfg"]["encoder"]["self_attention_model"]
    assert m.head.num_classes == plain["cfg"]["head"]["num_classes"]
    assert isinstance(m.decoding.tokenizer.vocab, list) and m.decoding.tokenizer.vocab == list(plain["cfg"]["decoding"]["vocabulary"])
    assert m.decoding.blank_id == len(m.decoding.tokenizer.vocab)
    assert M._plain(cfg.encoder) == plain["cfg"]["encoder"] and type(M._plain(cfg.decoding.vocabulary)) is list
    # every tensor of the state_dict (incl. the int64 counter and the torchaudio window) is staged for the library
    assert set(m._state) == set(ck["state_dict"])
    # load_model: named checkpoint under download_root, md5 gate, model_name written back into the cfg
    root = tmp_path / "cache"
    root.mkdir()
    path = root / "v2_ctc.ckpt"
    torch.save(ck, path)
    with pytest.raises(AssertionError, match="Model checksum failed"):
        gigaam_amd.load_model("v2_ctc", device="cpu", download_root=str(root))
    monkeypatch.setitem(gigaam_amd._MODEL_HASHES, "v2_ctc", hashlib.md5(path.read_bytes()).hexdigest())
    m2 = gigaam_amd.load_model("v2_ctc", device="cpu", download_root=str(root))
    assert m2.cfg.model_name == "v2_ctc" and m2._dtype == torch.float32        # fp16_encoder only applies on a GPU (reference :188)
    with pytest.raises(ValueError, match="not found"):
        gigaam_amd.load_model("v9_ctc", device="cpu", download_root=str(root))
    # Lightning-style fine-tuned checkpoint: base model by hyper_parameters.model_name, then its own weights on top
    ft = {"hyper_parameters": {"model_name": "v2_ctc"},
          "state_dict": {**{k: v + 0 for k, v in ck["state_dict"].items()}, "loss.weight": torch.zeros(1), "optimizer_junk": 3}}
    ftp = tmp_path / "finetuned.ckpt"
    torch.save(ft, ftp)
    m3 = gigaam_amd.load_model(str(ftp), device="cpu", download_root=str(root))
    assert isinstance(m3, gigaam_amd.GigaAMASR)
    assert set(m3._state) == {k for k in ck["state_dict"] if k.startswith(("preprocessor.", "encoder.", "head."))}


def test_rnnt_head_exposes_decoder_and_joint_views():
    """SURVEY 8b lists ``head.decoder`` / ``head.joint`` as attributes the reference reads (model.py:183-192,
    train_utils/module.py:130-144): present here as attribute views with the reference's names; their per-step entry
    points (r04: gam_rnnt_predict / gam_rnnt_joint) need the GPU like every compute call -- there is no CPU path."""
    from gigaam_amd.decoder import RNNTHead
    h = RNNTHead({"pred_hidden": 320, "pred_rnn_layers": 1, "num_classes": 34},
                 {"enc_hidden": 768, "pred_hidden": 320, "joint_hidden": 320, "num_classes": 34})
    assert (h.decoder.pred_hidden, h.decoder.pred_rnn_layers, h.decoder.num_classes, h.decoder.blank_id) == (320, 1, 34, 33)
    assert (h.joint.enc_hidden, h.joint.joint_hidden, h.joint.num_classes) == (768, 320, 34)
    from gigaam_amd._lib import GigaAMHipError
    for call in (lambda: h.decoder.predict(None, None), lambda: h.joint.joint(torch.zeros(1, 2, 768), torch.zeros(1, 1, 320))):
        with pytest.raises(GigaAMHipError, match="no CPU path|ROCm GPU"):
            call()
    with pytest.raises(AttributeError):
        h.joint.predict(None, None)
    assert "decoder" not in dict(h.named_children())       # views, not sub-modules: nothing to move or serialise


def test_gemm_plan_invariants():
    """gam_gemm_sp_plan is host code (no GPU): tile shape x split-K per launch from the time model fitted to
    profiles/r03_smallm_sweep.txt.  Pinned here: the invariants the kernel relies on (whole k-tiles per slice, at least four
    of them, split-K only for grids under half the chip) and the choices the sweep found best at the sizes that matter."""
    import ctypes as C
    from gigaam_amd import _lib
    lib = _lib.load_library()
    lib.gam_tune_sp(0, 0, 0)

    def plan(m, n, k, ncu=256):
        mt, nw, s = C.c_int(), C.c_int(), C.c_int()
        assert lib.gam_plan_sp(m, n, k, ncu, C.byref(mt), C.byref(nw), C.byref(s)) == 0
        return mt.value, nw.value, s.value

    for m in (5, 126, 501, 1004, 2008, 4016, 8032, 16064, 40000):
        for n, k in ((768, 768), (1536, 768), (3072, 768), (768, 3072), (768, 12288), (320, 768), (768, 6912)):
            mt, nw, s = plan(m, n, k)
            assert mt in (2, 3, 4) and nw in (2, 4) and not (nw == 2 and mt == 4) and 1 <= s <= 8
            nk = k // 32
            tiles = -(-m // (64 * mt)) * -(-n // (64 * nw))
            if s > 1:
                assert nk % s == 0 and nk // s >= 4 and tiles * 2 <= 256, (m, n, k, mt, nw, s)
    # the headline batch: full-width tiles, no split-K (sweep: 3x4 / 4x4 / 3x4 best)
    assert plan(16064, 768, 768) == (3, 4, 1) and plan(16064, 3072, 768) == (4, 4, 1) and plan(16064, 768, 3072) == (3, 4, 1)
    # a single clip and the 8-GPU strong point: split-K slices fill the chip
    assert plan(126, 768, 768)[2] >= 4 and plan(126, 768, 3072)[2] == 8 and plan(2008, 768, 3072)[2] >= 3
    assert plan(2008, 3072, 768)[2] == 1          # 384 tiles already: no slices
    # a partition with fewer CUs shifts the trade
    assert plan(4016, 768, 768, ncu=64)[2] == 1
    assert lib.gam_plan_sp(100, 100, 100, 256, C.byref(C.c_int()), C.byref(C.c_int()), C.byref(C.c_int())) != 0   # K % 32

    # r04: the fourth plan dimension -- LDS stages (profiles/r04_smallm_sweep_stages.txt).  Three stages exist for the
    # 4-wave tiles and the 128 x 256 tile only; the headline batch keeps the big two-stage tiles; forcing works both ways.
    def plan_ex(m, n, k):
        mt, nw, s, ns = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        assert lib.gam_plan_sp_ex(m, n, k, 256, C.byref(mt), C.byref(nw), C.byref(s), C.byref(ns)) == 0
        return mt.value, nw.value, s.value, ns.value

    assert lib.gam_tune_sp_stages(0) == 0
    for m in (5, 126, 1004, 2008, 4016, 8032, 16064):
        for n, k in ((768, 768), (1536, 768), (2304, 768), (3072, 768), (768, 3072)):
            mt, nw, s, ns = plan_ex(m, n, k)
            assert (mt, nw, s) == plan(m, n, k) and ns in (2, 3)
            if ns == 3:
                assert (nw == 2 and mt in (2, 3)) or (nw == 4 and mt == 2), (m, n, k, mt, nw)
    assert all(plan_ex(16064, n, k)[3] == 2 for n, k in ((768, 768), (3072, 768), (768, 3072), (2304, 768)))
    assert plan_ex(2008, 768, 768) == (2, 2, 2, 3) and plan_ex(126, 768, 768)[3] == 3      # the small grids: two k-tiles in flight
    assert lib.gam_tune_sp_stages(2) == 0 and plan_ex(2008, 768, 768)[3] == 2
    assert lib.gam_tune_sp_stages(3) == 0 and plan_ex(2008, 768, 768)[3] == 3
    assert lib.gam_tune_sp_stages(4) != 0
    assert lib.gam_tune_sp_stages(0) == 0


def test_decoded_object_carries_flag_and_event_explicitly():
    """r05: what a decode returns is ONE object whose hidden companions are explicit fields -- nothing hangs off
    a tensor view any more, so a slice / cat / unpack cannot silently drop the range flag.  CPU tensors: the plain collect path."""
    from gigaam_amd import shard
    from gigaam_amd.engine import Decoded, HipEngine
    ext = torch.tensor([2, 0, 3, 1], dtype=torch.int32)          # counts of 3 utterances + the flag word (set)
    ids = torch.arange(12, dtype=torch.int32).reshape(3, 4)
    frames = ids + 100
    dec = Decoded(ids, frames, ext[:3], ext=ext)
    a, b, c = dec                                                # unpacks like the old 3-tuple
    assert a is ids and b is frames and torch.equal(c, ext[:3]) and len(dec) == 3 and dec.counts is dec[2]
    assert not hasattr(c, "_gam_ext") and not hasattr(c, "_gam_evt")
    assert torch.equal(dec.flag_word(), ext[-1:]) and torch.equal(shard.range_flag_of(dec), ext[-1:])
    assert shard.range_flag_of(c) is None and shard.range_flag_of((ids, frames, c)) is None      # bare tensors know nothing
    rows, flag = HipEngine.collect(dec)
    assert flag is True and rows == [([0, 1], [100, 101]), ([], []), ([8, 9, 10], [108, 109, 110])]
    rows2, flag2 = HipEngine.collect(ids, frames, ext[:3])       # three bare tensors: same rows, flag unknown -> False
    assert rows2 == rows and flag2 is False
    d5 = Decoded(ids, frames, ext[:3], ext=ext, dump=torch.zeros(3, 2, 5), dump_count=torch.zeros(3, dtype=torch.int32))
    assert len(d5) == 5 and d5[:3][0] is ids
    # the decoding classes take the object whole
    from gigaam_amd.decoding import CTCGreedyDecoding, RangeOverflow
    d = CTCGreedyDecoding([chr(ord("a") + i) for i in range(12)])
    with pytest.raises(RangeOverflow):
        d.finish(dec)
    clean = Decoded(ids, frames, ext[:3], ext=torch.tensor([2, 0, 3, 0], dtype=torch.int32))
    assert [t for t, _, _ in d.finish(clean)] == ["ab", "", "ijk"]


def test_side_cluster_sizes():
    from gigaam_amd.engine import HipEngine
    assert [HipEngine.side_cluster(b, 96) for b in (1, 8, 9, 16, 32, 33, 64, 200)] == [8, 8, 6, 6, 3, 2, 1, 1]
    assert HipEngine.side_cluster(32, 160) == 5 and HipEngine.side_cluster(32, 32) == 1 and HipEngine.side_cluster(33, 8) == 1


def test_bench_mismatch_report_walks_to_the_first_differing_decision():
    """bench.py's cpu_baseline leg: for an utterance whose ids differ from the reference's, the report is the largest logit difference up to
    and including the first step whose argmax differs, and the reference's margin there."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    ref = torch.tensor([[0.0, -1.0, -2.0], [-0.5, -0.50001, -3.0], [-9.0, 0.0, -9.0]])
    gpu = ref.clone()
    gpu[0, 2] += 4e-4                      # inside the bar, same decision
    gpu[1, 1] += 3e-5                      # flips a near-tie
    gpu[2, 0] += 5.0                       # behind the divergence: must not count
    r = bench._mismatch_report(ref, gpu)
    assert r["first_differing_step"] == 1 and r["steps_compared"] == 2
    assert abs(r["max_logit_diff"] - 4e-4) < 1e-6 and abs(r["reference_margin_at_that_step"] - 1e-5) < 2e-6
    same = bench._mismatch_report(ref, ref.clone())
    assert same["first_differing_step"] is None and same["max_logit_diff"] == 0.0
