"""r06 diagnosis: v_pk_fma_f32 vs v_fmac_f32 on identical operands in a wave that shares its SIMD with another kernel (tools/pkfma_canary.hip).
    python tools/pkfma_canary.py [reps=6]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigaam_amd import synth  # noqa: E402
from gigaam_amd.engine import HipEngine, build_config  # noqa: E402

pk = C.CDLL(os.path.join(ROOT, "tools", "libpkfma_canary.so"))
pk.pkfma_canary_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
l1 = C.CDLL(os.path.join(ROOT, "tools", "libl1_canary.so"))
l1.l1c_aggressor_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 6
N_WG, SPIN_US = 256, 3000.0

cfg = synth.model_cfg("v2_ctc")
eng = HipEngine(build_config(cfg["preprocessor"], cfg["encoder"], None), {}, torch.device("cuda:0"))
g = torch.Generator().manual_seed(0)
w768 = (torch.randn(768, 768, generator=g) * 0.03).cuda()
x640 = torch.randn(640, 768, generator=g).cuda()
q = torch.randn(5, 120, 768, generator=g).cuda()
ql = torch.tensor([120, 100, 90, 77, 50]).cuda()
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
SRC_WORDS = 64 * 1024 * 1024
src = torch.randn(SRC_WORDS, device="cuda")
torch.cuda.synchronize()


def own(mode, n_wg=256):
    def fn():
        out = torch.zeros((n_wg, 4), dtype=torch.int32, device="cuda")
        assert l1.l1c_aggressor_launch(C.c_void_p(out.data_ptr()), C.c_void_p(src.data_ptr()), SRC_WORDS, mode, n_wg, 64 * 1024, SPIN_US,
                                       C.c_void_p(main.cuda_stream)) == 0
    return fn


def gemm640():
    for _ in range(110):
        eng.op_gemm(x640, w768)


def attention():
    for _ in range(110):
        eng.op_attention(q, q, q, ql)


AGG = [("none", lambda: None), ("own: MFMA loop", own(3)), ("own: MFMA loop x 512 workgroups", own(3, 512)), ("own: LDS-DMA (L2-hot)", own(1)),
       ("own: LDS-DMA + MFMA", own(5)), ("own: plain load stream", own(2)), ("own: LDS rw", own(4)), ("lib: op_gemm 640 (small-tile LDS-DMA GEMM)", gemm640),
       ("lib: op_attention", attention)]
MODES = ["registers", "h from LDS (broadcast b128)", "asm v_mov halves + op_sel", "hipcc-packed, weights as LDS quads", "hipcc-packed, weights as GLOBAL quads (counted vmcnt)"]
gw = torch.zeros(8 * 256 * 2 * 4, device="cuda")
print(f"reps {REPS} x {N_WG} workgroups x {SPIN_US / 1000:.0f} ms; chains of 32 FMAs, scalar v_fmac_f32 vs v_pk_fma_f32 on the same operands", flush=True)
for mode in (4, 3, 2, 0, 1):
    for aname, afn in AGG:
        lo = hi = iters = wgs = 0
        hist = np.zeros(8, dtype=np.int64)
        samples = []
        for _ in range(REPS):
            out = torch.zeros((N_WG, 32), dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                assert pk.pkfma_canary_launch(C.c_void_p(out.data_ptr()), mode, N_WG, 48 * 1024, SPIN_US, C.c_void_p(side.cuda_stream), C.c_void_p(gw.data_ptr())) == 0
            afn()
            torch.cuda.synchronize()
            o = out.cpu().numpy().view("uint32")
            lo += int(o[:, 0].sum()); hi += int(o[:, 1].sum()); iters += int(o[:, 2].sum()) * 256
            hit = (o[:, 0] + o[:, 1]) > 0
            wgs += int(hit.sum()); hist += o[:, 4:12].sum(axis=0).astype(np.int64)
            for r in o[hit][:2]:
                for k in range(int(r[12])):
                    a, b = np.uint32(r[20 + k]).view(np.float32), np.uint32(r[24 + k]).view(np.float32)
                    samples.append((int(r[16 + k]) & 0xffff, "lo" if r[16 + k] & 0x10000 else "hi", float(a), float(b), int(r[28 + k])))
        line = (f"mode {mode} [{MODES[mode]:54s}] beside {aname:44s}: chains {iters:.3e}  low-half mismatches {lo:8d}  high-half {hi:8d}  workgroups hit {wgs:5d} / {REPS * N_WG}")
        if lo + hi:
            line += f"\n      by 8-lane group {hist.tolist()}  samples (tid, half, scalar, packed, iteration) {samples[:4]}"
        print(line, flush=True)
