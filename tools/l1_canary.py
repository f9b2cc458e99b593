"""r06 diagnosis: do L1-HITTING global loads of one workgroup return wrong data while a co-resident workgroup of another kernel runs?
(tools/l1_canary.hip).  Stage 1: a 16 KB region re-read with plain loads beside every aggressor kind.  Stage 2: for every aggressor that
produced wrong words, region size x load flavour.
    python tools/l1_canary.py [reps=8]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigaam_amd import synth  # noqa: E402
from gigaam_amd.engine import HipEngine, build_config  # noqa: E402

lib = C.CDLL(os.path.join(ROOT, "tools", "libl1_canary.so"))
lib.l1c_fill.argtypes = [C.c_void_p, C.c_ulonglong, C.c_void_p]
lib.l1c_victim_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
lib.l1c_aggressor_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N_WG = 256
SPIN_US = 3000.0
INV = pow(2246822519, -1, 1 << 32)

cfg = synth.model_cfg("v2_ctc")
eng = HipEngine(build_config(cfg["preprocessor"], cfg["encoder"], None), {}, torch.device("cuda:0"))
g = torch.Generator().manual_seed(0)
w768 = (torch.randn(768, 768, generator=g) * 0.03).cuda()
x640 = torch.randn(640, 768, generator=g).cuda()
x16k = torch.randn(16064, 768, generator=g).cuda()
q = torch.randn(5, 120, 768, generator=g).cuda()
ql = torch.tensor([120, 100, 90, 77, 50]).cuda()
side = torch.cuda.Stream()
main = torch.cuda.current_stream()

VB_WORDS = 64 * 16384 * 4            # victim buffer: 64 regions of up to 256 KB
vbuf = torch.empty(VB_WORDS, dtype=torch.int32, device="cuda")
assert lib.l1c_fill(C.c_void_p(vbuf.data_ptr()), VB_WORDS, C.c_void_p(main.cuda_stream)) == 0
SRC_WORDS = 64 * 1024 * 1024         # aggressor source: 256 MB
src = torch.randn(SRC_WORDS, device="cuda")
torch.cuda.synchronize()


def aggr_own(mode, lds_kb=64, n_wg=256):
    def fn():
        out = torch.zeros((n_wg, 4), dtype=torch.int32, device="cuda")
        rc = lib.l1c_aggressor_launch(C.c_void_p(out.data_ptr()), C.c_void_p(src.data_ptr()), SRC_WORDS, mode, n_wg, lds_kb * 1024, SPIN_US,
                                      C.c_void_p(main.cuda_stream))
        assert rc == 0, rc
        return out
    return fn


def aggr_lib(kind):
    def fn():
        if kind == "gemm640":
            for _ in range(110):
                eng.op_gemm(x640, w768)
        elif kind == "gemm16k":
            for _ in range(40):
                eng.op_gemm(x16k, w768)
        elif kind == "attention":
            for _ in range(110):
                eng.op_attention(q, q, q, ql)
        return None
    return fn


AGGRESSORS = [("none", lambda: None), ("own: LDS-DMA stream (HBM)", aggr_own(0)), ("own: LDS-DMA stream (L2-hot)", aggr_own(1)),
              ("own: plain 16 B load stream", aggr_own(2)), ("own: MFMA loop", aggr_own(3)), ("own: LDS rw loop", aggr_own(4)),
              ("own: LDS-DMA + MFMA", aggr_own(5)), ("lib: op_gemm 640 x 768 x 768 (small-tile LDS-DMA)", aggr_lib("gemm640")),
              ("lib: op_gemm 16064 x 768 x 768", aggr_lib("gemm16k")), ("lib: op_attention", aggr_lib("attention"))]
FLAVOURS = ["plain", "nt", "sc1", "sc0 sc1"]


def cu_key(hw, xcc):
    return (int(xcc) & 0xf, (int(hw) >> 8) & 0xff)      # (XCC, SE/SH/CU field of HW_ID)


def run(aname, afn, region_kb, flavour, victim_lds_kb=48, reps=REPS):
    region_words = region_kb * 256
    bad_wg = bad = passes = 0
    samples, shared = [], []
    for _ in range(reps):
        out = torch.zeros((N_WG, 32), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            rc = lib.l1c_victim_launch(C.c_void_p(out.data_ptr()), C.c_void_p(vbuf.data_ptr()), region_words, 64, flavour, N_WG,
                                       victim_lds_kb * 1024, SPIN_US, C.c_void_p(side.cuda_stream))
            assert rc == 0, rc
        aout = afn()
        torch.cuda.synchronize()
        o = out.cpu().numpy().view("uint32")
        hit = o[:, 0] > 0
        bad_wg += int(hit.sum()); bad += int(o[:, 0].sum()); passes += int(o[:, 1].sum())
        if aout is not None:
            a = aout.cpu().numpy().view("uint32")
            vk = {cu_key(r[2], r[3]) for r in o}
            ak = {cu_key(r[0], r[1]) for r in a}
            shared.append(len(vk & ak) / max(1, len(vk)))
        for r in o[hit][:3]:
            for k in range(int(r[4])):
                idx, got, want = int(r[8 + k]), int(r[16 + k]), int(r[24 + k])
                prov = ((got ^ 0xA5000000) * INV) % (1 << 32)
                samples.append((idx, hex(got), hex(want), f"data of word {prov} (delta {prov - idx})" if prov < VB_WORDS else "not a pattern word"))
    msg = (f"{aname:52s} region {region_kb:4d} KB {FLAVOURS[flavour]:8s}: workgroups with a wrong word {bad_wg:5d} of {reps * N_WG}, wrong words {bad:7d}, "
           f"load batches {passes * 256}")
    if shared:
        msg += f", CUs shared with the aggressor {100 * np.mean(shared):.0f} %"
    if samples:
        msg += f"\n      samples (word index, got, want, provenance): {samples[:6]}"
    print(msg, flush=True)
    return bad


print(f"reps {REPS} x {N_WG} victim workgroups x {SPIN_US / 1000:.0f} ms; victim LDS 48 KB, own aggressors 64 KB of LDS", flush=True)
print("--- stage 1: 16 KB region (L1-resident), plain loads, every aggressor", flush=True)
hot = []
for aname, afn in AGGRESSORS:
    if run(aname, afn, 16, 0) > 0:
        hot.append((aname, afn))
print("--- stage 1b: 16 MB-like region (256 KB per workgroup: L1 misses), plain loads", flush=True)
for aname, afn in AGGRESSORS[1:3] + AGGRESSORS[7:8]:
    run(aname, afn, 256, 0)
print("--- stage 2: aggressors with wrong words:", [a for a, _ in hot], flush=True)
for aname, afn in hot[:3]:
    for region_kb in (4, 16, 24, 64):
        run(aname, afn, region_kb, 0)
    for flavour in (1, 2, 3):
        run(aname, afn, 16, flavour)
