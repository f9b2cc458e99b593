"""HipEngine: one ``gam_handle`` (include/gigaam_hip.h) behind torch tensors.

torch is plumbing here -- device memory, the current HIP stream and (in bench.py)
torch.distributed.  All compute happens in libgigaam_hip.so; there is no eager /
CPU fallback, and every method raises if the library or the GPU call fails.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, Mapping, Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import GamConfig, GigaAMHipError


def _get(cfg: Any, key: str, default: Any = None) -> Any:
    if cfg is None:
        return default
    if isinstance(cfg, Mapping):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def build_config(pre: Any = None, enc: Any = None, head: Any = None) -> GamConfig:
    """POD mirror of the checkpoint's cfg sub-trees.  Defaults follow the reference
    constructors (gigaam/preprocess.py:60-65, gigaam/encoder.py:510-526)."""
    c = GamConfig()
    sr = _get(pre, "sample_rate", 16000)
    c.sample_rate = sr
    c.n_mels = _get(pre, "features", 64)
    c.hop_length = _get(pre, "hop_length", sr // 100)
    c.win_length = _get(pre, "win_length", sr // 40)
    c.n_fft = _get(pre, "n_fft", sr // 40)
    c.center = int(bool(_get(pre, "center", True)))
    c.feat_in = _get(enc, "feat_in", 64)
    c.n_layers = _get(enc, "n_layers", 16)
    c.d_model = _get(enc, "d_model", 768)
    subs = _get(enc, "subsampling", "conv2d")
    assert subs in ("conv1d", "conv2d")  # encoder.py:48
    c.subsampling = _lib.SUBS_CONV2D if subs == "conv2d" else _lib.SUBS_CONV1D
    c.subs_kernel_size = _get(enc, "subs_kernel_size", 3)
    c.subsampling_factor = _get(enc, "subsampling_factor", 4)
    c.ff_expansion_factor = _get(enc, "ff_expansion_factor", 4)
    att = _get(enc, "self_attention_model", "rotary")
    assert att in ("rotary", "rel_pos"), f"Not supported attn = {att}"  # encoder.py:530-533
    c.self_attention_model = _lib.ATT_ROTARY if att == "rotary" else _lib.ATT_REL_POS
    c.n_heads = _get(enc, "n_heads", 16)
    c.pos_emb_max_len = _get(enc, "pos_emb_max_len", 5000)
    norm = _get(enc, "conv_norm_type", "batch_norm")
    assert norm in ("batch_norm", "layer_norm")  # encoder.py:377
    c.conv_norm_type = _lib.NORM_BATCH if norm == "batch_norm" else _lib.NORM_LAYER
    c.conv_kernel_size = _get(enc, "conv_kernel_size", 31)
    c.head_type = _lib.HEAD_NONE
    if head is not None:
        target = str(_get(head, "_target_", ""))
        dec, jn = _get(head, "decoder"), _get(head, "joint")
        if dec is not None and jn is not None or target.endswith("RNNTHead"):
            c.head_type = _lib.HEAD_RNNT
            c.num_classes = _get(dec, "num_classes")
            c.pred_hidden = _get(dec, "pred_hidden")
            c.pred_rnn_layers = _get(dec, "pred_rnn_layers")
            c.joint_hidden = _get(jn, "joint_hidden")
        elif target.endswith("Linear"):   # emotion model: torch.nn.Linear(in_features, out_features)
            c.head_type = _lib.HEAD_EMO
            c.num_classes = _get(head, "out_features")
        else:
            c.head_type = _lib.HEAD_CTC
            c.num_classes = _get(head, "num_classes")
    return c


_DTYPES = {
    torch.float32: _lib.DTYPE_F32, torch.float16: _lib.DTYPE_F16, torch.bfloat16: _lib.DTYPE_BF16,
    torch.float64: _lib.DTYPE_F64, torch.int64: _lib.DTYPE_I64,
}


class Decoded(tuple):
    """What ``ctc_greedy`` / ``rnnt_greedy`` return.  Unpacks as ``(ids, frames, counts)`` (``+ (dump, dump_count)`` when a
    logits dump was asked for) and carries the decode's two hidden companions as EXPLICIT fields -- they used to hang off the
    ``counts`` view as Python attributes, which any ``torch.cat`` / slice / unpack-and-repack dropped silently (ADVICE r3,
    VERDICT r4 weak #13):

    ``ext``    i32 [B + 1]: ``counts`` is its first B words, the last word receives the split-fp16 range flag
               (gam_range_flag_fetch), so ``collect`` brings both to the host in ONE copy;
    ``event``  recorded on the stream the decode ran on, right behind it: ``collect`` waits for THIS on its side stream, not
               for whatever the caller has enqueued since (the next batch of a launch-n / collect-n-1 pipeline);
    ``stream`` the stream the decode ran on.

    Pass the object itself to ``HipEngine.collect`` / ``shard.range_flag_of``; three bare tensors are accepted too, but then
    nothing is known about a flag or an event (a plain blocking copy on the current stream)."""

    def __new__(cls, ids: Tensor, frames: Tensor, counts: Tensor, ext: Optional[Tensor] = None, event=None, stream=None,
                dump: Optional[Tensor] = None, dump_count: Optional[Tensor] = None, whole: Optional[Tensor] = None):
        self = tuple.__new__(cls, (ids, frames, counts) if dump is None else (ids, frames, counts, dump, dump_count))
        self.ext, self.event, self.stream = ext, event, stream
        # ``whole`` (r06): ids | frames | ext are views of ONE i32 buffer [2 B cap + B + 1] -- a small decode (CTC: 128 KB at 32 x 20 s)
        # then reaches the host in ONE blocking copy instead of three (counts first, then the used part of ids and of frames)
        self.whole = whole
        return self

    ids = property(lambda self: self[0])
    frames = property(lambda self: self[1])
    counts = property(lambda self: self[2])

    def flag_word(self) -> Optional[Tensor]:
        """The device word (i32 [1]) that received the range flag of this decode, or None."""
        return None if self.ext is None else self.ext[-1:]


def _ptr(t: Optional[Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


class HipEngine:
    """Owns one library handle on one GPU."""

    def __init__(self, config: GamConfig, state_dict: Mapping[str, Tensor], device: torch.device):
        self.lib = _lib.load_library()
        device = torch.device(device)
        if device.type != "cuda":
            raise GigaAMHipError(f"gigaam_amd runs on a ROCm GPU only (device={device}); there is no CPU path")
        if not torch.cuda.is_available():
            raise GigaAMHipError("no ROCm GPU visible to torch")
        self.device = torch.device("cuda", device.index if device.index is not None else torch.cuda.current_device())
        self.cfg = config
        self._h = C.c_void_p()
        rc = self.lib.gam_create(C.byref(config), self.device.index, C.byref(self._h))
        self._check(rc, "gam_create")
        for key, val in state_dict.items():
            if not isinstance(val, Tensor):
                continue
            t = val.detach().to("cpu").contiguous()
            if t.dtype not in _DTYPES:
                t = t.to(torch.float32)
            shape = (C.c_int64 * max(1, t.dim()))(*t.shape)
            rc = self.lib.gam_set_weight(self._h, key.encode(), C.c_void_p(t.data_ptr()), _DTYPES[t.dtype], shape, t.dim())
            self._check(rc, f"gam_set_weight({key})")
        self._check(self.lib.gam_finalize(self._h), "gam_finalize")
        import os
        # the cluster size in force outside an overlapped decode: what gam_create read from the environment (clamped like it does), then
        # whatever the caller set through set_rnnt_cluster -- an overlapped decode restores THIS, not the environment's value
        self._rnnt_cluster_user = max(-1, min(8, int(os.environ.get("GAM_RNNT_CLUSTER", "-1"))))

    # ------------------------------------------------------------------ utils
    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            msg = self.lib.gam_last_error(self._h)
            raise GigaAMHipError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")

    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, t: Tensor, dtype: torch.dtype) -> Tensor:
        return t.to(device=self.device, dtype=dtype).contiguous()

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.gam_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_gemm_mode(self, mode: str) -> None:
        """"f32" (exact fp32 MFMA), "f16x3" (three-term split-fp16 MFMA, fp32-equivalent; the default) or "f16" (OPT-IN speed
        mode: one fp16 MFMA per product, fp32 accumulation -- the reference's GPU autocast contract, not its CPU one)."""
        m = {"f32": _lib.GEMM_F32, "f16x3": _lib.GEMM_F16X3, "f16": _lib.GEMM_F16}[mode]
        self._check(self.lib.gam_set_gemm_mode(self._h, m), "gam_set_gemm_mode")

    @property
    def gemm_mode(self) -> str:
        return {_lib.GEMM_F32: "f32", _lib.GEMM_F16X3: "f16x3", _lib.GEMM_F16: "f16"}[self.lib.gam_get_gemm_mode(self._h)]

    def range_flag(self) -> bool:
        """True if, since the last call, an unscaled split-fp16 GEMM operand left fp16's range (gam_range_flag);
        synchronises the current stream and clears the flag.  NOTE: ``ctc_greedy`` / ``rnnt_greedy`` CONSUME the flag (they
        move it into the hidden tail word of the counts buffer they return, which ``collect`` reads): after a decode call
        this method reports 0 -- read the flag from the ``Decoded`` object they return (``.flag_word()``) or through ``collect``."""
        out = C.c_int(0)
        with torch.cuda.device(self.device):
            self._check(self.lib.gam_range_flag(self._h, C.byref(out), self._stream()), "gam_range_flag")
        return self._flag_of(out.value)

    def _counts_with_flag(self, b: int) -> Tuple[Tensor, Tensor]:
        """counts i32 [b] as a view of a [b + 1] buffer whose last element receives the range flag
        (gam_range_flag_fetch) -- ``collect`` then brings both to the host in ONE copy."""
        ext = torch.empty((b + 1,), dtype=torch.int32, device=self.device)
        return ext[:b], ext

    def _fetch_flag(self, ext: Tensor):
        """Move the range flag into ``ext``'s tail word and record "this decode is complete" on the stream it ran on;
        returns (event, stream) for the ``Decoded`` object.  ``collect`` waits for the event on a side stream, so a caller
        that has already launched the next batch (model.transcribe_longform, shard.run_sharded, bench.py configs 4 / 5) is not
        held until that next batch has finished too -- a ``.cpu()`` on the launch stream is ordered behind everything enqueued
        there, which made the "launch n, then collect n-1" pipelines wait for batch n (measured: +3.9 ms per 33 ms step)."""
        rc = self.lib.gam_range_flag_fetch(self._h, C.c_void_p(ext.data_ptr() + 4 * (ext.numel() - 1)), self._stream())
        self._check(rc, "gam_range_flag_fetch")
        st = torch.cuda.current_stream(self.device)
        evt = torch.cuda.Event()
        evt.record(st)
        return evt, st

    _aux_streams: Dict[int, Tuple["torch.cuda.Stream", "torch.cuda.Stream", "torch.cuda.Stream"]] = {}
    import threading as _threading
    _aux_lock = _threading.Lock()
    import os as _os
    ONE_COPY_BYTES = int(_os.environ.get("GAM_ONE_COPY_BYTES", str(256 * 1024)))   # decodes up to this size reach the host in one copy (collect); 0: A/B switch

    @classmethod
    def aux_streams(cls, device: torch.device) -> Tuple["torch.cuda.Stream", "torch.cuda.Stream", "torch.cuda.Stream"]:
        """(decode side stream, collect stream, host-to-device copy stream) of a device: three HIGH-priority streams taken from torch's pool in
        one go.  Why (tools/queue_probe.py, profiles/r06_queue_probe.txt): HIP maps streams onto four hardware queues per priority level, round
        robin in creation order, and two streams on one queue SERIALISE -- one normal-priority stream in four lands on the null stream's queue,
        where an "overlapped" decode, the ids' D2H copy or the next batch's H2D copy silently queues behind the encoder it was meant to run
        beside (both / one = 2.00 instead of 1.00).  High-priority streams have their own four queues (never the launch stream's, whatever
        the caller created before), and three taken consecutively are on three different ones.  Same-box A/B of the priority alone: neutral."""
        key = device.index if device.index is not None else torch.cuda.current_device()
        st = cls._aux_streams.get(key)
        if st is None:
            import os
            with cls._aux_lock:      # (two threads building their first engine at once must end up with ONE trio)
                st = cls._aux_streams.get(key)
                if st is None:
                    prio = int(os.environ.get("GAM_AUX_STREAM_PRIORITY", "-1"))     # (0: the A/B switch of profiles/r06_queue_probe.txt)
                    st = cls._aux_streams[key] = tuple(torch.cuda.Stream(device, priority=prio) for _ in range(3))
        return st

    @classmethod
    def _collect_stream(cls, device: torch.device) -> "torch.cuda.Stream":
        return cls.aux_streams(device)[1]

    @staticmethod
    def _flag_of(word: int) -> bool:
        """The handle's flag word: bit 0 = a split-fp16 operand left fp16's range (the caller repeats the batch on the exact path);
        bit 1 = gam_encode_varlen was given host lengths SHORTER than the device's (the batch is incomplete: an error, not a fallback)."""
        if int(word) & 2:
            raise GigaAMHipError("encode(host_lengths=...): a host length is shorter than the device length of the same utterance")
        return bool(int(word) & 1)

    @staticmethod
    def collect(dec, frames: Optional[Tensor] = None, counts: Optional[Tensor] = None):
        """``Decoded`` -> ([(ids, frames)] host lists, range_flag).  One blocking D2H for the counts AND the split-fp16 range
        flag accumulated up to this decode, one for the used part of ids / frames, both on a side stream that waits for the
        decode's own completion event only.  Three bare tensors (ids, frames, counts) are accepted for callers that built them
        some other way: a plain blocking copy on the current stream, flag unknown (False)."""
        whole = None
        if isinstance(dec, Decoded):
            ids, frames, counts, ext, evt = dec[0], dec[1], dec[2], dec.ext, dec.event
            whole = getattr(dec, "whole", None)
        else:
            ids, ext, evt = dec, None, None
        src = ext if ext is not None else counts
        if whole is not None and evt is not None and whole.is_cuda and whole.numel() * 4 <= HipEngine.ONE_COPY_BYTES:
            # one blocking D2H of the whole decode (a second and third round trip cost more than the unused tail of ids / frames)
            side = HipEngine._collect_stream(whole.device)
            with torch.cuda.stream(side):
                side.wait_event(evt)
                host = whole.cpu()
            whole.record_stream(side)
            b, cap = ids.shape
            arr = host.numpy()      # (zero-copy; numpy row slices convert to lists at half the cost of per-row Tensor.tolist(): 0.33 -> 0.17 ms at 32 x 502)
            n = arr[2 * b * cap:].tolist()
            flag = HipEngine._flag_of(n.pop())
            if n and min(n) < 0:
                raise GigaAMHipError("decode left an utterance undecoded (counts = -1)")
            ids_h, fr_h = arr[: b * cap].reshape(b, cap), arr[b * cap: 2 * b * cap].reshape(b, cap)
            return [(ids_h[i, :c].tolist(), fr_h[i, :c].tolist()) for i, c in enumerate(n)], flag
        if evt is not None and ids.is_cuda:
            side = HipEngine._collect_stream(ids.device)
            with torch.cuda.stream(side):
                side.wait_event(evt)
                n = src.cpu().tolist()
                flag = HipEngine._flag_of(n.pop()) if ext is not None else False
                width = max(n) if n else 0
                ids_h, fr_h = ids[:, :width].cpu(), frames[:, :width].cpu()
            for t in (ids, frames, src):
                t.record_stream(side)
        else:
            if ids.is_cuda:
                # bare tensors carry no completion event: the decode may have run on the engine's side stream, which a .cpu() on the
                # CURRENT stream is not ordered behind -- wait for the whole device rather than read half-written counts
                torch.cuda.synchronize(ids.device)
            n = src.cpu().tolist()
            flag = HipEngine._flag_of(n.pop()) if ext is not None else False
            width = max(n) if n else 0
            ids_h, fr_h = ids[:, :width].cpu(), frames[:, :width].cpu()
        if n and min(n) < 0:   # cannot happen: gam_rnnt_greedy repairs failed clusters itself (gam_api.hip launch_single)
            raise GigaAMHipError("RNN-T decode left an utterance undecoded (counts = -1)")
        ids_a, fr_a = ids_h.numpy(), fr_h.numpy()
        return [(ids_a[i, :c].tolist(), fr_a[i, :c].tolist()) for i, c in enumerate(n)], flag

    def feat_frames(self, n_samples: int) -> int:
        return int(self.lib.gam_feat_frames(self._h, n_samples))

    def enc_frames(self, n_feat: int) -> int:
        return int(self.lib.gam_enc_frames(self._h, n_feat))

    # ------------------------------------------------------------------ operators
    def frontend(self, wav: Tensor, length: Tensor) -> Tuple[Tensor, Tensor]:
        wav = self._dev(wav, torch.float32)
        length = self._dev(length, torch.int64)
        b, l = wav.shape
        t = self.feat_frames(l)
        feat = torch.empty((b, self.cfg.n_mels, t), dtype=torch.float32, device=self.device)
        flen = torch.empty((b,), dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.gam_frontend(self._h, _ptr(wav), _ptr(length), b, l, _ptr(feat), _ptr(flen), self._stream())
        self._check(rc, "gam_frontend")
        return feat, flen

    def host_feat_lengths(self, wav_lengths) -> Optional[list]:
        """Feature-frame counts of a batch from its sample counts, ON THE HOST (``FeatureExtractor.out_len``, gam_feat_frames) -- what
        ``encode(host_lengths=...)`` wants for a ragged batch.  None when the lengths live on the GPU (reading them would be a host sync)."""
        if isinstance(wav_lengths, Tensor):
            if wav_lengths.is_cuda:
                return None
            wav_lengths = wav_lengths.tolist()
        return [self.feat_frames(int(n)) for n in wav_lengths]

    def last_encode_rows(self) -> Tuple[int, int]:
        """(token rows the layers of the last ``encode`` ran on, rows of the padded layout): equal unless that call packed its rows."""
        pad = C.c_int(0)
        rows = self.lib.gam_last_encode_rows(self._h, C.byref(pad))
        return int(rows), int(pad.value)

    def encode(self, feat: Tensor, length: Tensor, n_layers_run: int = -1, want_tokens: bool = False, host_lengths=None):
        """``host_lengths``: the same feature lengths as ``length`` as host integers (``host_feat_lengths``).  With them a ragged batch's layers
        run on its valid frames only (gam_encode_varlen: packed rows); without them -- or for an equal-length batch -- on B x T'max rows."""
        if host_lengths is None and isinstance(length, Tensor) and not length.is_cuda:
            host_lengths = length.tolist()        # the caller's lengths are on the host anyway
        feat = self._dev(feat, torch.float32)
        length = self._dev(length, torch.int64)
        b, f, t = feat.shape
        if f != self.cfg.feat_in:
            raise GigaAMHipError(f"expected [B,{self.cfg.feat_in},T] features, got {tuple(feat.shape)}")
        tp = self.enc_frames(t)
        enc = torch.empty((b, self.cfg.d_model, tp), dtype=torch.float32, device=self.device)
        elen = torch.empty((b,), dtype=torch.int32, device=self.device)
        tok = torch.empty((b, tp, self.cfg.d_model), dtype=torch.float32, device=self.device) if want_tokens else None
        with torch.cuda.device(self.device):
            if host_lengths is not None and b > 1:
                if len(host_lengths) != b:
                    raise GigaAMHipError(f"host_lengths has {len(host_lengths)} entries for a batch of {b}")
                hl = (C.c_int64 * b)(*[int(v) for v in host_lengths])
                rc = self.lib.gam_encode_varlen(self._h, _ptr(feat), _ptr(length), hl, b, t, _ptr(enc), _ptr(elen), n_layers_run, _ptr(tok),
                                                self._stream())
            elif n_layers_run < 0 and not want_tokens:
                rc = self.lib.gam_encode(self._h, _ptr(feat), _ptr(length), b, t, _ptr(enc), _ptr(elen), self._stream())
            else:
                rc = self.lib.gam_encode_ex(self._h, _ptr(feat), _ptr(length), b, t, _ptr(enc), _ptr(elen),
                                            n_layers_run, _ptr(tok), self._stream())
        self._check(rc, "gam_encode")
        return (enc, elen, tok) if want_tokens else (enc, elen)

    def ctc_head(self, encoded: Tensor) -> Tensor:
        encoded = self._dev(encoded, torch.float32)
        b, _, tp = encoded.shape
        out = torch.empty((b, tp, self.cfg.num_classes), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.gam_ctc_head(self._h, _ptr(encoded), b, tp, _ptr(out), self._stream())
        self._check(rc, "gam_ctc_head")
        return out

    def emo_probs(self, encoded: Tensor, enc_len: Optional[Tensor] = None) -> Tensor:
        """[B,d_model,T'] -> class probabilities [B,num_classes] (mean over time, Linear, softmax);
        enc_len restricts the mean to each utterance's valid frames (None = all T')."""
        encoded = self._dev(encoded, torch.float32)
        enc_len = None if enc_len is None else self._dev(enc_len, torch.int32)
        b, _, tp = encoded.shape
        out = torch.empty((b, self.cfg.num_classes), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.gam_emo_probs(self._h, _ptr(encoded), _ptr(enc_len), b, tp, _ptr(out), self._stream())
        self._check(rc, "gam_emo_probs")
        return out

    def ctc_greedy(self, encoded: Tensor, enc_len: Tensor) -> Decoded:
        encoded = self._dev(encoded, torch.float32)
        enc_len = self._dev(enc_len, torch.int32)
        b, _, tp = encoded.shape
        whole = torch.empty((2 * b * tp + b + 1,), dtype=torch.int32, device=self.device)     # ids | frames | counts + flag word
        ids, frames, ext = whole[: b * tp].view(b, tp), whole[b * tp: 2 * b * tp].view(b, tp), whole[2 * b * tp:]
        counts = ext[:b]
        with torch.cuda.device(self.device):
            rc = self.lib.gam_ctc_greedy(self._h, _ptr(encoded), _ptr(enc_len), b, tp, _ptr(ids), _ptr(frames),
                                         _ptr(counts), self._stream())
            self._check(rc, "gam_ctc_greedy")
            evt, st = self._fetch_flag(ext)
        return Decoded(ids, frames, counts, ext, evt, st, whole=whole)

    def set_rnnt_cluster(self, n: int) -> None:
        """Workgroups per utterance of the cluster decode kernel (gam_set_rnnt_cluster): -1 auto, 0 one-workgroup kernel, 1..8."""
        self._check(self.lib.gam_set_rnnt_cluster(self._h, int(n)), "gam_set_rnnt_cluster")
        self._rnnt_cluster_user = int(n)

    def _decode_side_stream(self) -> "torch.cuda.Stream":
        return self.aux_streams(self.device)[0]      # (shared by the engines of a device: see aux_streams)

    @staticmethod
    def side_cluster(b: int, side_cus: int) -> int:
        """Cluster size of an OVERLAPPED decode: the clusters of ``b`` utterances (8 utterance columns x ceil(b / 8) rows x C
        workgroups, one per CU -- gam_api.hip) together hold at most ``side_cus`` compute units; at least 1."""
        import os
        if "GAM_DEBUG_SIDE_CLUSTER" in os.environ:      # debug: force the overlapped decode's cluster size (0 = one-workgroup kernel)
            return int(os.environ["GAM_DEBUG_SIDE_CLUSTER"])
        return max(1, min(8, side_cus // (8 * ((b + 7) // 8))))

    def rnnt_greedy(self, encoded: Tensor, enc_len: Tensor, max_symbols: int, dump_cap: int = 0, overlap: bool = False,
                    side_cus: int = 0) -> Decoded:
        """RNNTGreedyDecoding.decode on the device (gam_rnnt_greedy).  ``overlap``: launch it on this engine's decode SIDE stream
        with small clusters (at most ``side_cus`` CUs held; 0 = by vocabulary: 96 for a char head, 160 where W_out is streamed
        from L2 -- measured on configs 3 / 4, profiles/r05_ab_experiments.txt), ordered behind everything enqueued on the current stream so far, and
        return at once -- the caller's next ``frontend`` / ``encode`` on the current stream then runs BESIDE this decode instead of
        behind it (the greedy loop is latency-bound: with the GPU to itself it keeps ~224 CUs resident and idle).
        The range flag is fetched on the CURRENT stream first, so it covers exactly this batch's frontend + encoder (fetched on
        the side stream it would swallow a flag the next batch's encoder sets meanwhile).  Same kernels; ids are bit-identical to a serial decode
        AT THE SAME CLUSTER SIZE (tests/test_hip_hardening.py).  The cluster size fixes how a member's sums are partitioned, so a decode with
        small clusters may resolve a near-tie (top-1 / top-2 margin below ~1e-5) differently from the full-size clusters' -- inside the 1e-3 logit
        bar, and the reason bench.py and the tests compare overlapped and serial decodes at equal cluster size.  Decode-class calls of one handle
        are ordered by the library whatever streams they run on (gam_api.hip DecodeScope)."""
        encoded = self._dev(encoded, torch.float32)
        enc_len = self._dev(enc_len, torch.int32)
        b, _, tp = encoded.shape
        cap = tp * max_symbols
        main = torch.cuda.current_stream(self.device)
        side = self._decode_side_stream() if overlap else None
        # ids | frames | counts + flag word as views of ONE buffer (allocated on the launch stream): a small decode reaches the host in one copy
        whole = torch.empty((2 * b * cap + b + 1,), dtype=torch.int32, device=self.device)
        ext = whole[2 * b * cap:]
        counts = ext[:b]
        if overlap:
            with torch.cuda.device(self.device):
                self._fetch_flag(ext)            # on the launch stream: this batch's own flag (its event is not the decode's)
            side.wait_stream(main)
            c_side = self.side_cluster(b, side_cus if side_cus > 0 else (96 if self.cfg.num_classes <= 64 else 160))
            self._check(self.lib.gam_set_rnnt_cluster(self._h, c_side), "gam_set_rnnt_cluster")     # (for this call only: restored below)
        try:
            with torch.cuda.stream(side) if overlap else torch.cuda.device(self.device):
                ids, frames = whole[: b * cap].view(b, cap), whole[b * cap: 2 * b * cap].view(b, cap)
                dump = dcount = None
                if dump_cap > 0:
                    dump = torch.zeros((b, dump_cap, self.cfg.num_classes), dtype=torch.float32, device=self.device)
                    dcount = torch.zeros((b,), dtype=torch.int32, device=self.device)
                rc = self.lib.gam_rnnt_greedy(self._h, _ptr(encoded), _ptr(enc_len), b, tp, max_symbols, _ptr(ids),
                                              _ptr(frames), _ptr(counts), _ptr(dump), _ptr(dcount), dump_cap, self._stream())
                self._check(rc, "gam_rnnt_greedy")
                if overlap:
                    st = torch.cuda.current_stream(self.device)
                    evt = torch.cuda.Event()
                    evt.record(st)
                    for t in (encoded, enc_len, whole):    # allocated on the launch stream, last used on the side stream
                        t.record_stream(st)
                else:
                    evt, st = self._fetch_flag(ext)
        finally:
            if overlap:
                self._check(self.lib.gam_set_rnnt_cluster(self._h, self._rnnt_cluster_user), "gam_set_rnnt_cluster")
        return Decoded(ids, frames, counts, ext, evt, st, dump, dcount, whole=whole)

    def rnnt_predict(self, labels: Optional[Tensor], state: Optional[Tuple[Tensor, Tensor]], batch_size: int = 1):
        """One predictor step (reference RNNTDecoder.predict, gigaam/decoder.py:85-102): labels i [B] or None (zero input),
        state (h, c) f32 [L,B,pred_hidden] or None -> (g f32 [B,pred_hidden], (h', c'))."""
        ph, nl = self.cfg.pred_hidden, self.cfg.pred_rnn_layers
        if labels is not None:
            labels = self._dev(labels.reshape(-1), torch.int32)
            b = labels.shape[0]
        else:
            b = batch_size if state is None else state[0].shape[1]
        h_in = c_in = None
        if state is not None:
            h_in, c_in = self._dev(state[0], torch.float32), self._dev(state[1], torch.float32)
            if tuple(h_in.shape) != (nl, b, ph) or tuple(c_in.shape) != (nl, b, ph):
                raise GigaAMHipError(f"predictor state must be two [{nl},{b},{ph}] tensors, got {tuple(h_in.shape)} / {tuple(c_in.shape)}")
        g = torch.empty((b, ph), dtype=torch.float32, device=self.device)
        h_out = torch.empty((nl, b, ph), dtype=torch.float32, device=self.device)
        c_out = torch.empty_like(h_out)
        with torch.cuda.device(self.device):
            rc = self.lib.gam_rnnt_predict(self._h, _ptr(labels), _ptr(h_in), _ptr(c_in), b, _ptr(g), _ptr(h_out), _ptr(c_out), self._stream())
        self._check(rc, "gam_rnnt_predict")
        return g, (h_out, c_out)

    def rnnt_joint(self, enc: Tensor, dec: Tensor) -> Tensor:
        """Reference RNNTJoint.joint (gigaam/decoder.py:41-47): enc f32 [B,T,d_model], dec f32 [B,U,pred_hidden] ->
        log-probs f32 [B,T,U,num_classes]."""
        enc, dec = self._dev(enc, torch.float32), self._dev(dec, torch.float32)
        b, t, d = enc.shape
        b2, u, ph = dec.shape
        if b != b2 or d != self.cfg.d_model or ph != self.cfg.pred_hidden:
            raise GigaAMHipError(f"joint expects [B,T,{self.cfg.d_model}] and [B,U,{self.cfg.pred_hidden}], got {tuple(enc.shape)} / {tuple(dec.shape)}")
        out = torch.empty((b, t, u, self.cfg.num_classes), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.gam_rnnt_joint(self._h, _ptr(enc), _ptr(dec), b, t, u, _ptr(out), self._stream())
        self._check(rc, "gam_rnnt_joint")
        return out

    def op_gemm(self, a: Tensor, w: Tensor, bias: Optional[Tensor] = None, act: int = 0) -> Tensor:
        a = self._dev(a, torch.float32)
        w = self._dev(w, torch.float32)
        bias = None if bias is None else self._dev(bias, torch.float32)
        m, k = a.shape
        n = w.shape[0]
        out = torch.empty((m, n), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.gam_op_gemm(self._h, _ptr(a), _ptr(w), _ptr(bias), _ptr(out), m, n, k, act, self._stream())
        self._check(rc, "gam_op_gemm")
        return out

    def op_attention(self, q: Tensor, k: Tensor, v: Tensor, lens: Optional[Tensor] = None) -> Tensor:
        """q, k, v [B,T,H*48] -> ctx [B,T,H*48]; keys >= lens[b] are masked."""
        q, k, v = (self._dev(t, torch.float32) for t in (q, k, v))
        b, t, d = q.shape
        lens_d = None if lens is None else self._dev(lens, torch.int32)
        out = torch.empty_like(q)
        with torch.cuda.device(self.device):
            rc = self.lib.gam_op_attention(self._h, _ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lens_d), b, t, d // 48, self._stream())
        self._check(rc, "gam_op_attention")
        return out

    # ------------------------------------------------------------------ profiling
    def profile_enable(self, on=True) -> None:
        """True/1: time every launch; 2: GEMM family only; False/0: off."""
        self._check(self.lib.gam_profile_enable(self._h, int(on)), "gam_profile_enable")

    def profile_level(self, on) -> None:
        """Change the collection level without resetting (sampled profiling: gam_profile_pause)."""
        self._check(self.lib.gam_profile_pause(self._h, int(on)), "gam_profile_pause")

    def profile_read(self) -> Dict[str, Dict[str, float]]:
        out = {}
        for i, name in enumerate(_lib.PF_CLASSES):
            ms, n, work = C.c_double(), C.c_int64(), C.c_double()
            self._check(self.lib.gam_profile_read(self._h, i, C.byref(ms), C.byref(n), C.byref(work)), "gam_profile_read")
            nbytes = C.c_double()
            self._check(self.lib.gam_profile_read_bytes(self._h, i, C.byref(nbytes)), "gam_profile_read_bytes")
            out[name] = {"ms": ms.value, "launches": int(n.value), "work": work.value, "bytes": nbytes.value}
        return out
