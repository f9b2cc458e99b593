"""Debug 3 (r05): is the exact-fp32 GEMM family (gam_gemm.h: the heads' GEMMs) deterministic when an encoder runs on another stream?"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gigaam_amd
from gigaam_amd import synth

ck = synth.make_checkpoint("v2_rnnt", seed=1, n_layers=2, rnnt_blank_bias=13.5)
model = gigaam_amd.model_from_checkpoint(ck, "cuda:0")
eng = model.encoder.engine
REPS = int(sys.argv[1])
lens = [int(16000 * (1.0 + 0.37 * ((3 * i + 1) % 11))) for i in range(32)]
w, l = synth.synth_audio(32, max(lens) / 16000.0, seed=301, lengths=lens)
w, l = w.cuda(), l.cuda()
lens5 = [int(16000 * (1.0 + 0.37 * ((3 * i + 2) % 11))) for i in range(5)]
w5, l5 = synth.synth_audio(5, max(lens5) / 16000.0, seed=302, lengths=lens5)
w5, l5 = w5.cuda(), l5.cuda()
enc, elen = eng.encode(*eng.frontend(w, l))
tok = enc.transpose(1, 2).contiguous()                     # [B, T, D]
dec = torch.randn(32, 1, 320, device="cuda")
ref = eng.rnnt_joint(tok, dec).clone()
ref_gemm = eng.op_gemm(tok.reshape(-1, 768), torch.randn(320, 768, generator=torch.Generator().manual_seed(1))).clone()
wgt = torch.randn(320, 768, generator=torch.Generator().manual_seed(1)).cuda()
torch.cuda.synchronize()
side = torch.cuda.Stream()
bad_joint = bad_gemm = bad_serial = 0
for rep in range(REPS):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        o = eng.rnnt_joint(tok, dec)
        o2 = eng.op_gemm(tok.reshape(-1, 768), wgt)
    eng.encode(*eng.frontend(w5, l5))
    eng.encode(*eng.frontend(w5, l5))
    torch.cuda.synchronize()
    bad_joint += not torch.equal(o, ref)
    bad_gemm += not torch.equal(o2, ref_gemm)
    o3 = eng.rnnt_joint(tok, dec)
    torch.cuda.synchronize()
    bad_serial += not torch.equal(o3, ref)
print(f"reps {REPS}: joint (fp32 GEMMs + log-softmax) beside an encoder differs {bad_joint}x, op_gemm (split-fp16) {bad_gemm}x, joint alone {bad_serial}x")
