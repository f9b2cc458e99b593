import sys, os, torch, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import load_case, oracle_features, valid_mask
from gigaam_amd.engine import HipEngine, build_config
ck, wav, wlen, gold = load_case("v1_rnnt_l2")
feat_o, flen = oracle_features(ck, wav, wlen)
sd = ck["state_dict"]
win = sd["preprocessor.featurizer.0.spectrogram.window"].double(); fb = sd["preprocessor.featurizer.0.mel_scale.fb"]
spec = torch.stft(wav.double(), n_fft=400, hop_length=160, win_length=400, window=win, center=True, pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
f64 = torch.log(torch.matmul((spec.real**2+spec.imag**2).transpose(-1,-2), fb.double()).transpose(-1,-2).clamp(1e-9,1e9))
fm = valid_mask(feat_o.shape[2], flen)[:, None, :]
cfg = ck["cfg"]
eng = HipEngine(build_config(cfg["preprocessor"], cfg["encoder"], cfg.get("head")), ck["state_dict"], torch.device("cuda:0"))
for mode in ("f16x3", "f32"):
    eng.set_gemm_mode(mode)
    feat, _ = eng.frontend(wav, wlen)
    feat = feat.cpu()
    for db in (30, 50, 60, 80, 200):
        strong = f64 >= f64.max(dim=1, keepdim=True).values - db * 0.2302585
        d64 = (feat.double() - f64).abs() * fm * strong
        do = (feat - feat_o).abs() * fm * strong
        print(mode, db, "vs fp64 %.2e  vs oracle %.2e" % (float(d64.max()), float(do.max())))
    d = (feat.double() - f64).abs() * fm * (f64 >= f64.max(dim=1, keepdim=True).values - 60 * 0.2302585)
    b, m, t = torch.nonzero(d == d.max())[0].tolist()
    print("  worst strong at", b, m, t, "f64", float(f64[b, m, t]), "hip", float(feat[b, m, t]), "oracle", float(feat_o[b, m, t]), "frame max", float(f64[b, :, t].max()), "flen", flen.tolist())
