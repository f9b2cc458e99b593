"""CPU: the C-ABI library loads and exports every declared symbol; host-side logic."""
import os
import re
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
                assert "ref_shim" not in src, f


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
