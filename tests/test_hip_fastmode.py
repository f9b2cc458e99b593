"""GPU: the OPT-IN one-term fp16 speed mode (GAM_GEMM_F16, include/gigaam_hip.h) -- the arithmetic contract of the reference's own
GPU default (fp16 autocast + half() encoder, /root/reference/gigaam/model.py:34-37, gigaam/__init__.py:188-189), narrower than
the fp32 CPU reference every parity claim of this package is made against.  Tested for what it promises: fp16-product accuracy
(not fp32), the same control flow, never the default, and the default mode's results untouched by a round trip through it."""
import numpy as np
import pytest
import torch

from common import load_case, ragged_from_device, report, split_ragged, valid_mask

pytestmark = pytest.mark.gpu

# fp16 products: 11 significant bits per operand, fp32 accumulation.  A length-K dot product of O(1) terms is off by
# ~2^-11 sqrt(K) relative to its terms; through two Conformer layers (LayerNorm-scaled activations) the golden cases measure
# 2e-3 .. 6e-3 on the encoder output (reported in $GAM_TEST_REPORT).
TOL_GEMM_REL = 2e-3
TOL_ENC_F16 = 3e-2
TOL_ATT_F16 = 5e-3


def _engine(ck, sp=True):
    import os
    from gigaam_amd.engine import HipEngine, build_config
    cfg = ck["cfg"]
    old = os.environ.get("GAM_SP_MIN_M")
    os.environ["GAM_SP_MIN_M"] = "1" if sp else str(1 << 30)
    try:
        return HipEngine(build_config(cfg["preprocessor"], cfg["encoder"], cfg.get("head")), ck["state_dict"], torch.device("cuda:0"))
    finally:
        if old is None:
            del os.environ["GAM_SP_MIN_M"]
        else:
            os.environ["GAM_SP_MIN_M"] = old


def test_default_mode_is_not_the_speed_mode():
    ck, _, _, _ = load_case("v2_ctc_l2")
    eng = _engine(ck)
    assert eng.gemm_mode == "f16x3"
    eng.set_gemm_mode("f16")
    assert eng.gemm_mode == "f16"
    eng.set_gemm_mode("f16x3")
    assert eng.gemm_mode == "f16x3"
    with pytest.raises(KeyError):
        eng.set_gemm_mode("fp8")


@pytest.mark.parametrize("shape", [(2500, 768, 768, 0), (300, 3072, 768, 1), (4016, 768, 3072, 0), (16064, 768, 768, 2), (126, 1536, 768, 0), (2008, 3072, 768, 1)])
def test_gemm_one_term(shape):
    """gam_op_gemm under GAM_GEMM_F16 against fp64: within fp16-product accuracy, and measurably NOT the three-term result
    (the one-term kernels really ran); every tile class of the one-term build is reached by the shapes."""
    m, n, k, act = shape
    ck, _, _, _ = load_case("v2_ctc_l2")
    eng = _engine(ck)
    g = torch.Generator().manual_seed(m + n)
    a = torch.randn(m, k, generator=g).cuda()
    w = (torch.randn(n, k, generator=g) / k ** 0.5).cuda()
    b = torch.randn(n, generator=g).cuda()
    ref = a.double() @ w.double().t() + b.double()
    if act == 1:
        ref = ref * torch.sigmoid(ref)
    if act == 2:
        ref = ref.clamp_min(0)
    scale = float(ref.abs().max())
    err3 = float((eng.op_gemm(a, w, b, act).double() - ref).abs().max()) / scale
    eng.set_gemm_mode("f16")
    err1 = float((eng.op_gemm(a, w, b, act).double() - ref).abs().max()) / scale
    eng.set_gemm_mode("f16x3")
    err3b = float((eng.op_gemm(a, w, b, act).double() - ref).abs().max()) / scale
    report("gemm_one_term", shape=list(shape), rel_err_f16=err1, rel_err_f16x3=err3)
    assert err1 < TOL_GEMM_REL, (shape, err1)
    assert err1 > 20 * err3, (shape, err1, err3)          # three terms keep 22 bits, one keeps 11
    assert err3b == err3                                   # the default mode is bit-for-bit what it was before the round trip


def test_attention_one_term():
    ck, _, _, _ = load_case("v2_ctc_l2")
    eng = _engine(ck)
    g = torch.Generator().manual_seed(3)
    B, T, H = 3, 300, 16
    q, k, v = (torch.randn(B, T, H * 48, generator=g) for _ in range(3))
    lens = torch.tensor([300, 211, 64], dtype=torch.int32)
    qh, kh, vh = (t.view(B, T, H, 48).transpose(1, 2).double() for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) / 48 ** 0.5
    s = s.masked_fill(~valid_mask(T, lens)[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, T, H * 48)
    qm = valid_mask(T, lens)[:, :, None]
    e3 = float(((eng.op_attention(q, k, v, lens).cpu().double() - ref) * qm).abs().max())
    eng.set_gemm_mode("f16")
    e1 = float(((eng.op_attention(q, k, v, lens).cpu().double() - ref) * qm).abs().max())
    report("attention_one_term", err_f16=e1, err_f16x3=e3)
    assert e1 < TOL_ATT_F16 and e1 > 20 * e3, (e1, e3)


@pytest.mark.parametrize("case", ["v2_ctc_l2", "v3_ctc_l2", "v1_ctc_l2", "v3_e2e_rnnt_l2"])
def test_encoder_in_speed_mode(case):
    """The whole encoder under GAM_GEMM_F16 against the reference's fp32 golden: fp16-product accuracy on the activations
    (measured 2e-3 .. 6e-3), identical lengths, and a decode that runs (ids are compared and REPORTED, not asserted: the golden
    margins, 2e-3, were chosen for fp32 arithmetic)."""
    ck, wav, wlen, gold = load_case(case)
    eng = _engine(ck)
    eng.set_gemm_mode("f16")
    enc, elen = eng.encode(*eng.frontend(wav, wlen))
    assert elen.cpu().tolist() == gold["enc_len"].tolist()
    vm = valid_mask(enc.shape[2], gold["enc_len"])[:, None, :]
    err = float(((enc.cpu() - torch.from_numpy(gold["encoded"])) * vm).abs().max())
    ms = ck["cfg"]["decoding"].get("max_symbols_per_step", 10)
    out = eng.rnnt_greedy(enc, elen, ms) if "rnnt" in case else eng.ctc_greedy(enc, elen)
    # the decode call CONSUMED the engine's range flag (it rides in the Decoded object's tail word): read it from there --
    # eng.range_flag() after a decode always reports 0 and could not see an fp16 overflow of the one-term mode
    from gigaam_amd.engine import HipEngine
    got, flag = HipEngine.collect(out)
    assert not flag
    ref = split_ragged(gold["ids"], gold["frames"], gold["counts"].tolist())
    same = sum(a == b for a, b in zip(got, ref))
    report("encoder_speed_mode", case=case, err=err, tol=TOL_ENC_F16, utterances_identical=f"{same}/{len(ref)}")
    assert 1e-4 < err < TOL_ENC_F16, err        # (> 1e-4: this IS the narrower arithmetic, not the default mode by mistake)
    assert all(len(i) == len(f) for i, f in got)


@pytest.mark.parametrize("case", ["v2_ctc_l2", "v1_ctc_l2", "v3_ctc_l2"])
def test_speed_mode_on_a_large_grid(case):
    """The kernels a LARGE grid takes in the speed mode -- 128-query attention workgroups (rotary and relative-position), the
    8-wave GEMM tiles, no split-K -- on 6 x 30 s through the two-layer models: close to the default mode's output (which the
    long-utterance parity tests pin to the oracle), same lengths, no range flag."""
    from gigaam_amd import synth
    ck, _, _, _ = load_case(case)
    eng = _engine(ck)
    wav, wlen = synth.synth_audio(6, 30.0, seed=11, lengths=[480000, 470000, 300000, 480000, 123456, 480000])
    feat, flen = eng.frontend(wav, wlen)
    enc3, elen3 = eng.encode(feat, flen)
    eng.set_gemm_mode("f16")
    enc1, elen1 = eng.encode(*eng.frontend(wav, wlen))
    assert elen1.cpu().tolist() == elen3.cpu().tolist()
    vm = valid_mask(enc3.shape[2], elen3.cpu())[:, None, :].to(enc3.device)
    err = float(((enc1 - enc3) * vm).abs().max())
    report("speed_mode_large_grid", case=case, err_vs_default=err, tol=TOL_ENC_F16)
    assert 1e-5 < err < TOL_ENC_F16, err
    assert not eng.range_flag()
    out = ragged_from_device(*eng.ctc_greedy(enc1, elen1))
    assert all(len(i) == len(f) and (not f or f[-1] < n) for (i, f), n in zip(out, elen1.cpu().tolist()))
