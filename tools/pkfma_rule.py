"""r06 diagnosis: which instruction pattern makes v_pk_fma_f32 unreliable beside another kernel's MFMAs (tools/pkfma_rule.hip)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rl = C.CDLL(os.path.join(ROOT, "tools", "libpkfma_rule.so"))
rl.pkfma_rule_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
l1 = C.CDLL(os.path.join(ROOT, "tools", "libl1_canary.so"))
l1.l1c_aggressor_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N_WG, SPIN_US = 256, 3000.0
TESTS = ["control: pair written 8 states earlier, no modifiers", "LOW half of src0 written immediately in front", "HIGH half of src0 written immediately in front",
         "as 1 + s_nop 0", "as 1 + s_nop 1", "as 1 + s_nop 3", "nothing fresh, op_sel_hi:[1,0,1]", "nothing fresh, op_sel:[0,1,0]",
         "write-after-read: src0 LOW overwritten right behind", "write-after-read: src0 HIGH overwritten right behind",
         "LOW half of src1 written immediately in front", "LOW half of src2 (accumulator) written immediately in front"]
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
src = torch.randn(1 << 20, device="cuda")
torch.cuda.synchronize()
for beside in ("MFMA loop", "none"):
    for t, name in enumerate(TESTS):
        lo = hi = it = g = 0
        for _ in range(REPS):
            out = torch.zeros((N_WG, 4), dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                assert rl.pkfma_rule_launch(C.c_void_p(out.data_ptr()), t, N_WG, SPIN_US, C.c_void_p(side.cuda_stream)) == 0
            if beside != "none":
                ao = torch.zeros((256, 4), dtype=torch.int32, device="cuda")
                assert l1.l1c_aggressor_launch(C.c_void_p(ao.data_ptr()), C.c_void_p(src.data_ptr()), 1 << 20, 3, 256, 64 * 1024, SPIN_US, C.c_void_p(main.cuda_stream)) == 0
            torch.cuda.synchronize()
            o = out.cpu().numpy().view("uint32")
            lo += int(o[:, 0].sum()); hi += int(o[:, 1].sum()); it += int(o[:, 2].sum()) * 256; g += int(o[:, 3].sum())
        print(f"beside {beside:9s} test {t:2d} [{name:62s}]: chains {it:.3e}  low-half mismatches {lo:9d}  high-half {hi:9d}  of them in lanes 48..63 {g:9d}", flush=True)

rl.pkfma_single_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
SINGLE = {12: "v_pk_fma_f32 op_sel:[1,0,0] (low result <- src0 HIGH)", 13: "v_pk_fma_f32 op_sel:[0,0,1] (low result <- src2 HIGH)",
          14: "v_pk_mul_f32 op_sel:[0,1]", 15: "v_pk_add_f32 op_sel:[0,1]", 16: "v_pk_fma_f32 op_sel:[0,1,0] (low result <- src1 HIGH)",
          17: "v_fma_mix_f32 op_sel:[0,1,0] (src1 = fp16 high half)", 18: "v_pk_fma_f16 op_sel:[0,1,0]", 19: "v_pk_mul_f32 op_sel_hi:[1,0]",
          20: "v_pk_fma_f32 op_sel:[0,1,0], MFMAs issued by waves 4..7 of the SAME workgroup"}
for beside in ("MFMA loop", "none"):
    for t, name in SINGLE.items():
        self_mfma = 1 if t == 20 else 0
        if self_mfma and beside != "none":
            continue
        lo = hi = it = g = 0
        for _ in range(REPS):
            out = torch.zeros((N_WG, 4), dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                assert rl.pkfma_single_launch(C.c_void_p(out.data_ptr()), t, N_WG, 512 if self_mfma else 256, self_mfma, SPIN_US, C.c_void_p(side.cuda_stream)) == 0
            if beside != "none":
                ao = torch.zeros((256, 4), dtype=torch.int32, device="cuda")
                assert l1.l1c_aggressor_launch(C.c_void_p(ao.data_ptr()), C.c_void_p(src.data_ptr()), 1 << 20, 3, 256, 64 * 1024, SPIN_US, C.c_void_p(main.cuda_stream)) == 0
            torch.cuda.synchronize()
            o = out.cpu().numpy().view("uint32")
            lo += int(o[:, 0].sum()); hi += int(o[:, 1].sum()); it += int(o[:, 2].sum()) * 256; g += int(o[:, 3].sum())
        print(f"beside {beside:9s} test {t:2d} [{name:78s}]: instructions {it:.3e}  low-result mismatches {lo:9d}  high-result {hi:9d}  of them in lanes 48..63 {g:9d}", flush=True)
