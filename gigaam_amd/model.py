"""Model assembly with the reference's public surface (gigaam/model.py:16-259):
``GigaAM.forward / embed_audio / prepare_wav``, ``GigaAMASR.transcribe /
transcribe_longform``.  The four cfg slots are instantiated by class path exactly
like ``hydra.utils.instantiate`` does in the reference (model.py:24-25,93-94), but
resolve to the HIP-backed operators of this package, which share ONE library
handle per model.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Mapping, Optional, Sequence, Tuple

import torch
from torch import Tensor, nn

from . import decoder as _decoder
from . import decoding as _decoding
from . import encoder as _encoder
from . import preprocess as _preprocess
from .engine import HipEngine, build_config
from .preprocess import SAMPLE_RATE, load_audio
from .types import LongformTranscriptionResult, Segment, TranscriptionResult, Word

LONGFORM_THRESHOLD = 25 * SAMPLE_RATE

_TARGETS = {
    "FeatureExtractor": _preprocess.FeatureExtractor,
    "ConformerEncoder": _encoder.ConformerEncoder,
    "CTCHead": _decoder.CTCHead,
    "RNNTHead": _decoder.RNNTHead,
    "Linear": _decoder.EmoHead,   # cfg.head of the emotion model (torch.nn.Linear)
    "CTCGreedyDecoding": _decoding.CTCGreedyDecoding,
    "RNNTGreedyDecoding": _decoding.RNNTGreedyDecoding,
}


def _node(cfg: Any, key: str) -> Any:
    return cfg[key] if isinstance(cfg, Mapping) else getattr(cfg, key)


def _plain(x: Any) -> Any:
    """A cfg node as plain dict / list / scalar.  The reference's checkpoints hold omegaconf containers
    (``DictConfig`` is a ``MutableMapping``, ``ListConfig`` a ``Sequence``, both also attribute-style); anything with
    the mapping or sequence PROTOCOL is accepted, by interface, not by class name."""
    if isinstance(x, Mapping) or (hasattr(x, "keys") and hasattr(x, "__getitem__")):
        return {k: _plain(x[k]) for k in x.keys()}
    if isinstance(x, (str, bytes)):
        return x
    if isinstance(x, Sequence):
        return [_plain(v) for v in x]
    return x


def instantiate(node: Any) -> Any:
    """``_target_`` class-path instantiation; 'gigaam.encoder.ConformerEncoder' and
    'gigaam_amd.encoder.ConformerEncoder' both resolve here."""
    kw = dict(_plain(node))
    target = kw.pop("_target_")
    cls = _TARGETS.get(target.rsplit(".", 1)[-1])
    if cls is None:
        raise ValueError(f"no MI355X implementation for {target}")
    return cls(**kw)


class GigaAM(nn.Module):
    _check_range = True   # forward() reads the split-fp16 range flag (one 4-byte D2H + sync) and falls back to fp32

    def __init__(self, cfg: Any):
        super().__init__()
        self.cfg = cfg
        self.preprocessor = instantiate(_node(cfg, "preprocessor"))
        self.encoder = instantiate(_node(cfg, "encoder"))

    def _head_cfg(self) -> Any:
        return None

    def load_state_dict(self, state_dict: Mapping[str, Tensor], strict: bool = True, assign: bool = False):  # type: ignore[override]
        """Hand every tensor to the library once (re-laid out there) and share the handle."""
        self._state = {k: v for k, v in state_dict.items() if isinstance(v, Tensor)}
        self._build_engine()
        return nn.modules.module._IncompatibleKeys([], [])

    def _build_engine(self) -> None:
        dev = self._device
        if dev.type != "cuda" or getattr(self, "_state", None) is None:
            return  # built on .to(device)
        conf = build_config(_node(self.cfg, "preprocessor"), _node(self.cfg, "encoder"), self._head_cfg())
        eng = HipEngine(conf, self._state, dev)
        self._state = None
        for m in (self.preprocessor, self.encoder, getattr(self, "head", None)):
            if m is not None:
                m.attach(eng)

    def _apply(self, fn, recurse=True):  # .to(device) / .cuda()
        out = super()._apply(fn, recurse)
        self._build_engine()
        return out

    def forward(self, features: Tensor, feature_lengths: Tensor) -> Tuple[Tensor, Tensor]:
        """wav [B,L], len [B] -> encoded [B,d_model,T'], len i32 [B]  (model.py:27-37;
        the reference wraps the encoder in fp16 autocast on GPU, this path stays fp32).  The public call checks the
        split-fp16 range flag itself (one 4-byte D2H + sync: the caller is about to read the tensor anyway); the
        transcribe paths read the flag together with the decode counts instead (``_encode`` + decoding.finish)."""
        host = self._host_feat_lengths(feature_lengths)     # (lengths given on the CPU: a ragged batch runs on its valid frames only)
        features, feature_lengths = self.preprocessor(features, feature_lengths)
        out = self.encoder(features, feature_lengths, host) if host is not None else self.encoder(features, feature_lengths)
        eng = getattr(self.encoder, "engine", None)
        if self._check_range and eng is not None and eng.gemm_mode != "f32" and eng.range_flag():
            # an activation outside fp16's range reached a split-fp16 GEMM operand (include/gigaam_hip.h,
            # gam_range_flag): the batch is repeated on the exact-fp32 MFMA path, which has no such limit
            self._warn_range("this batch was")
            mode = eng.gemm_mode
            eng.set_gemm_mode("f32")
            try:
                out = self.encoder(features, feature_lengths, host) if host is not None else self.encoder(features, feature_lengths)
            finally:
                eng.set_gemm_mode(mode)
        return out

    def _host_feat_lengths(self, wav_lengths) -> Optional[list]:
        """Feature-frame counts on the host when the caller's sample counts are on the host (no sync is ever made to get them)."""
        eng = getattr(self.encoder, "engine", None)
        if eng is None or not hasattr(eng, "host_feat_lengths"):
            return None
        return eng.host_feat_lengths(wav_lengths)

    def set_arithmetic(self, mode: str) -> None:
        """Arithmetic of the dense contractions (include/gigaam_hip.h, gam_set_gemm_mode): "f16x3" -- the default, a
        three-term fp16 split with fp32 accumulation, fp32-equivalent (what every parity claim of this package is made in);
        "f32" -- exact fp32 MFMA; "f16" -- OPT-IN speed mode, one fp16 MFMA per product: the contract of the reference's
        own GPU default (fp16 autocast, gigaam/model.py:34-37), not of its CPU path.  Not part of the reference's API."""
        self.encoder.engine.set_gemm_mode(mode)

    @staticmethod
    def _warn_range(what: str) -> None:
        import warnings
        warnings.warn(f"gigaam_amd: activation beyond the split-fp16 GEMM range; {what} recomputed with "
                      "GAM_GEMM_F32 (set GAM_GEMM_MODE=f32 to use that path throughout)", RuntimeWarning, stacklevel=3)

    def _encode(self, wav: Tensor, lengths: Tensor, host_lengths=None) -> Tuple[Tensor, Tensor]:
        """Internal twin of ``forward`` for the transcribe paths: fp32 in, fp32 out whatever ``fp16_encoder`` says (the
        fp16 storage contract of reference __init__.py:188-189 applies to what ``forward`` / ``embed_audio`` RETURN, not
        to what the head consumes here), no host sync and no range check -- the caller reads the flag with the counts."""
        host = self._host_feat_lengths(lengths if host_lengths is None else host_lengths)
        feat, flen = self.preprocessor(wav.to(torch.float32), lengths)
        return self.encoder.forward_f32(feat, flen, host)

    def _with_f32_fallback(self, fn: Callable[[], Any], what: str) -> Any:
        """Run ``fn``; if the decode it collects reports the range flag (decoding.RangeOverflow), warn and run it again
        under GAM_GEMM_F32."""
        from .decoding import RangeOverflow
        try:
            return fn()
        except RangeOverflow:
            eng = self.encoder.engine
            self._warn_range(what)
            mode = eng.gemm_mode
            eng.set_gemm_mode("f32")
            try:
                return fn()
            finally:
                eng.set_gemm_mode(mode)

    @property
    def _device(self) -> torch.device:
        return next(self.parameters()).device

    @property
    def _dtype(self) -> torch.dtype:
        # the reference's first parameter is the encoder's (its FeatureExtractor holds buffers only): model.py:43-45
        return next(self.encoder.parameters()).dtype

    def prepare_wav(self, wav_file: str) -> Tuple[Tensor, Tensor]:
        wav = load_audio(wav_file)
        wav = wav.to(self._device).to(self._dtype).unsqueeze(0)
        length = torch.full([1], wav.shape[-1], device=self._device)
        return wav, length

    def embed_audio(self, wav_file: str) -> Tuple[Tensor, Tensor]:
        wav, length = self.prepare_wav(wav_file)
        return self.forward(wav, length)

    def _prepare_wav_f32(self, wav_file: str) -> Tuple[Tensor, Tensor]:
        """``prepare_wav`` without the cast to ``_dtype``: the transcribe paths feed the fp32 frontend fp32 samples
        (PCM16 / 32768 is exact in fp32; through fp16 it would keep 11 bits)."""
        wav = load_audio(wav_file).to(self._device).unsqueeze(0)
        return wav, torch.full([1], wav.shape[-1], device=self._device)

    def _encode_checked(self, wav: Tensor, lengths: Tensor) -> Tuple[Tensor, Tensor]:
        """``_encode`` + the blocking range check of ``forward`` (for heads that bring no counts to the host)."""
        out = self._encode(wav, lengths)
        eng = self.encoder.engine
        if self._check_range and eng.gemm_mode != "f32" and eng.range_flag():
            self._warn_range("this batch was")
            mode = eng.gemm_mode
            eng.set_gemm_mode("f32")
            try:
                out = self._encode(wav, lengths)
            finally:
                eng.set_gemm_mode(mode)
        return out


class GigaAMEmo(GigaAM):
    """Emotion recognition model (reference gigaam/model.py:262-285): encoder + time pooling +
    linear head + softmax; ``cfg.id2name`` maps class index -> name."""

    def __init__(self, cfg: Any):
        super().__init__(cfg)
        self.head = instantiate(_node(cfg, "head"))
        self.id2name = _plain(_node(cfg, "id2name"))

    def _head_cfg(self) -> Any:
        return _node(self.cfg, "head")

    def get_probs(self, wav_file: str) -> Dict[str, float]:
        wav, length = self._prepare_wav_f32(wav_file)
        encoded, _ = self._encode_checked(wav, length)
        # the reference pools over the whole T' axis of its single, unpadded file (model.py:278-280)
        probs = self.head.probs(encoded)[0].tolist()
        names = self.id2name
        return {(names[i] if not isinstance(names, dict) else names.get(i, names.get(str(i)))): probs[i] for i in range(len(probs))}

    def get_probs_batch(self, wav: Tensor, lengths: Tensor) -> Tensor:
        """Batched variant: wav [B,L], lengths [B] -> probabilities [B,n_classes]; the mean runs over each
        utterance's valid encoder frames."""
        encoded, enc_len = self._encode_checked(wav.to(self._device), lengths.to(self._device))
        return self.head.probs(encoded, enc_len)


class GigaAMASR(GigaAM):
    def __init__(self, cfg: Any):
        super().__init__(cfg)
        self.head = instantiate(_node(cfg, "head"))
        self.decoding = instantiate(_node(cfg, "decoding"))

    def _head_cfg(self) -> Any:
        return _node(self.cfg, "head")

    def _with_words(self, decoded, wav_lens: Tensor, encoded_len: Tensor, word_timestamps: bool, event=None):
        """``event``: the decode's completion event (``engine.Decoded.event``) when the caller has one -- the lengths are then
        copied on the collect stream behind THAT event; without one the collect stream waits for the current stream (correct
        for any caller, e.g. the reference-style ``_decode(enc_cuda, enc_len_cuda, wav_lens_cpu, word_timestamps=True)``)."""
        if not word_timestamps:
            return [(text, None) for text, _, _ in decoded]
        from .timestamps_utils import compute_frame_shift, frames_to_words

        on_dev = [t for t in (wav_lens, encoded_len) if t.is_cuda]
        if on_dev:
            # on the collect stream: a copy on the launch stream would queue behind the NEXT batch's kernels in the one-batch
            # pipelines.  The ordering is explicit: the decode's own event, or everything enqueued so far
            dev = on_dev[0].device
            side = HipEngine._collect_stream(dev)
            if event is not None:
                side.wait_event(event)
            else:
                side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                wl, el = wav_lens.cpu().tolist(), encoded_len.cpu().tolist()
            for t in on_dev:          # (record_stream is not defined for host tensors)
                t.record_stream(side)
        else:
            wl, el = wav_lens.tolist(), encoded_len.tolist()
        out: List[Tuple[str, Optional[List[Word]]]] = []
        for i, (text, ids, frames) in enumerate(decoded):
            shift = compute_frame_shift(int(wl[i]), int(el[i]))
            out.append((text, frames_to_words(self.decoding.tokenizer, ids, frames, shift)))
        return out

    def _decode(self, encoded: Tensor, encoded_len: Tensor, wav_lens: Tensor,
                word_timestamps: bool = False) -> List[Tuple[str, Optional[List[Word]]]]:
        """reference model.py:96-124"""
        return self._with_words(self.decoding.decode(self.head, encoded, encoded_len), wav_lens, encoded_len, word_timestamps)

    # ---- the launch / collect pair every transcribe path is built from (public: a driver -- bench.py, shard.run_sharded,
    #      a test -- can interleave them, or replace them to script the decode)
    def launch_batch(self, wav: Tensor, lengths: Tensor, overlap: bool = False, host_lengths=None):
        """Device half of ``transcribe_batch``: frontend + encoder + greedy decode of a collated batch (wav [B,L] zero
        padded, len [B]) launched on the current stream, NO host sync.  Returns an opaque handle for ``collect_batch``.
        ``overlap=True`` says another ``launch_batch`` follows before this one is collected: an RNN-T decode then runs on
        the decode side stream BESIDE the next batch's encoder (decoding.RNNTGreedyDecoding.decode_device)."""
        # sample counts on the host (given, or ``lengths`` itself still a CPU tensor): a ragged batch then runs on its valid frames only
        host = host_lengths if host_lengths is not None else (lengths if (isinstance(lengths, Tensor) and not lengths.is_cuda) else None)
        wav, lengths = wav.to(self._device), lengths.to(self._device)
        encoded, encoded_len = self._encode(wav, lengths, host)
        return self.decoding.decode_device(self.head, encoded, encoded_len, overlap=overlap), lengths, encoded_len

    def collect_batch(self, handle, word_timestamps: bool = False) -> List[Tuple[str, Optional[List[Word]]]]:
        """Host half: blocks on the handle's decode, detokenises, builds word timestamps.  Raises
        ``decoding.RangeOverflow`` if the split-fp16 range flag was set (the callers below repeat in fp32)."""
        dev_out, wav_lens, encoded_len = handle
        return self._with_words(self.decoding.finish(dev_out), wav_lens, encoded_len, word_timestamps, event=getattr(dev_out, "event", None))

    @torch.inference_mode()
    def transcribe(self, wav_file: str, word_timestamps: bool = False) -> TranscriptionResult:
        wav, length = self._prepare_wav_f32(wav_file)
        if length.item() > LONGFORM_THRESHOLD:
            raise ValueError("Too long wav file, use 'transcribe_longform' method.")
        text, words = self.transcribe_batch(wav, length, word_timestamps)[0]
        return TranscriptionResult(text=text, words=words)

    @torch.inference_mode()
    def transcribe_batch(self, wav: Tensor, lengths: Tensor, word_timestamps: bool = False):
        """Batched twin of ``transcribe`` on an already collated batch (wav [B,L] zero
        padded, len [B]) -- the unit bench.py and the longform driver iterate."""
        return self._with_f32_fallback(lambda: self.collect_batch(self.launch_batch(wav, lengths), word_timestamps), "this batch was")

    @torch.inference_mode()
    def transcribe_longform(self, wav_file: str, word_timestamps: bool = False, fr_batch_size: int = 16,
                            fr_num_workers: int = 0, **kwargs: Any) -> LongformTranscriptionResult:
        """Segment -> zero-padded batches of ``fr_batch_size`` -> transcribe -> stitch
        (reference model.py:195-259).  The reference segments with pyannote's VAD
        (gated third-party model, not installable here); pass ``speech_regions=[(s,e),..]``,
        ``vad=callable(wav, sr) -> regions`` or ``vad="energy"`` (vad_utils.EnergyVAD, a labelled
        stand-in) and the reference's own chunk packer
        (vad_utils.pack_regions) does the rest.

        ``fr_num_workers`` is accepted for signature compatibility and ignored: the reference hands it to a
        ``DataLoader`` whose workers only run ``collate`` (model.py:219-229); here that role is the pinned,
        double-buffered ``feeder.BatchFeeder`` (one staging thread is enough to keep the GPU busy)."""
        from .vad_utils import EnergyVAD, segment_audio_file

        if kwargs.get("vad") == "energy":   # stand-in detector on the HIP frontend (NOT pyannote)
            kwargs["vad"] = EnergyVAD(self.preprocessor)
        elif kwargs.get("vad") is None and kwargs.get("speech_regions") is None:
            # reference default: pyannote (model.py:212-216 -> vad_utils.py:100-101).  Without that optional
            # third-party package the call still works, on the labelled stand-in, and says so.
            try:
                import pyannote.audio  # noqa: F401
            except ImportError:
                import warnings
                warnings.warn("pyannote.audio is not installed: transcribe_longform() segments with gigaam_amd's "
                              "EnergyVAD stand-in instead of the reference's pyannote/segmentation-3.0 pipeline "
                              "(chunk boundaries will differ); pass speech_regions= or vad= to control segmentation",
                              RuntimeWarning, stacklevel=2)
                kwargs["vad"] = EnergyVAD(self.preprocessor)
        segments, boundaries = segment_audio_file(wav_file, SAMPLE_RATE, device=self._device, **kwargs)
        if not segments:
            return LongformTranscriptionResult(segments=[])
        from .feeder import BatchFeeder

        # One-batch software pipeline: the kernels of batch n are launched (launch_batch: no host sync) BEFORE the
        # decoded ids of batch n-1 are copied back and detokenised (collect_batch), so the D2H wait, the tokenizer and
        # the feeder's staging of the next batch all run while the GPU works.  (The reference syncs per batch:
        # model.py:230-236.)  The range flag of the split-fp16 GEMMs rides on each batch's counts copy; if it ever
        # fires, the file is redone on the fp32 path.
        def run() -> List[Segment]:
            result: List[Segment] = []

            def emit(handle) -> None:
                for text, words in self.collect_batch(handle, word_timestamps):
                    start, end = boundaries[len(result)]
                    if word_timestamps:
                        shifted = [Word(text=w.text, start=round(w.start + start, 3), end=round(w.end + start, 3)) for w in words or []]
                        result.append(Segment(text=text, start=start, end=end, words=shifted))
                    else:
                        result.append(Segment(text=text, start=start, end=end))

            pending = None
            n_batches = (len(segments) + fr_batch_size - 1) // fr_batch_size
            feeder = BatchFeeder(segments, fr_batch_size, self._device)                           # pinned, double-buffered H2D
            for k, (wav, lens) in enumerate(feeder):
                handle = self.launch_batch(wav, lens, overlap=k + 1 < n_batches, host_lengths=getattr(feeder, "host_lengths", None))   # (RNN-T: decode n beside encoder n+1)
                if pending is not None:
                    emit(pending)
                pending = handle
            if pending is not None:
                emit(pending)
            return result

        eng = getattr(self.encoder, "_engine", None)
        if eng is not None and eng.gemm_mode != "f32":
            eng.range_flag()     # a flag left behind by earlier direct engine use must not cost this file an fp32 rerun
        return LongformTranscriptionResult(segments=self._with_f32_fallback(run, "the file was"))
