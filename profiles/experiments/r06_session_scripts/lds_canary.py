"""r05 diagnosis: which kernel writes into ANOTHER workgroup's LDS?  A canary kernel (tools/lds_canary.hip) holds a pattern in its LDS on a side
stream while the launch stream runs one kind of kernel of the library; any changed word was written by someone else.
    python tools/lds_canary.py [lds_kb=56] [reps=40]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigaam_amd import synth  # noqa: E402
from gigaam_amd.engine import HipEngine, build_config  # noqa: E402

lib = C.CDLL(os.path.join(ROOT, "tools", "liblds_canary.so"))
lib.lds_canary_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
LDS_KB = int(sys.argv[1]) if len(sys.argv) > 1 else 56
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 40
N_WG = 256
OUT_WORDS = 4 + 12 + 2

cfg = synth.model_cfg("v2_ctc")
eng = HipEngine(build_config(cfg["preprocessor"], cfg["encoder"], None), {}, torch.device("cuda:0"))
g = torch.Generator().manual_seed(0)
w768 = (torch.randn(768, 768, generator=g) * 0.03).cuda()
w3072 = (torch.randn(3072, 768, generator=g) * 0.03).cuda()
x640 = torch.randn(640, 768, generator=g).cuda()
x2008 = torch.randn(2008, 768, generator=g).cuda()
x16k = torch.randn(16064, 768, generator=g).cuda()
q = torch.randn(5, 120, 768, generator=g).cuda()
ql = torch.tensor([120, 100, 90, 77, 50]).cuda()
side = torch.cuda.Stream()


def run(name, fn, spin_us=3000.0, lds_kb=LDS_KB):
    tot_changed = tot_wg = 0
    lo, hi, samples = 1 << 30, 0, []
    for _ in range(REPS):
        out = torch.zeros((N_WG, OUT_WORDS), dtype=torch.int32, device="cuda")
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            rc = lib.lds_canary_launch(C.c_void_p(out.data_ptr()), N_WG, lds_kb * 1024, spin_us, C.c_void_p(side.cuda_stream))
            assert rc == 0, rc
        fn()
        torch.cuda.synchronize()
        o = out.cpu().numpy().view("uint32")
        hit = o[:, 0] > 0
        tot_changed += int(o[:, 0].sum()); tot_wg += int(hit.sum())
        if hit.any():
            lo = min(lo, int(o[hit, 1].min())); hi = max(hi, int(o[hit, 2].max()))
            for r in o[hit][:2]:
                samples.append([(int(r[4 + k]), hex(int(r[10 + k]))) for k in range(int(r[3]))])
    print(f"{name:44s} canary LDS {lds_kb:3d} KB: workgroups hit {tot_wg:5d} of {REPS * N_WG}, words changed {tot_changed:7d}"
          + (f", byte offsets {lo} .. {hi}, samples {samples[:2]}" if tot_wg else ""), flush=True)


def run_late(name, fn, lds_kb=LDS_KB, n_canaries=12, spin_us=150.0):
    """The canaries START while the other stream's kernels are already running: their workgroups are placed on LDS that a retiring
    workgroup of the other kernel has just freed -- a write of that kernel still in flight at its retirement would land in the canary."""
    tot_changed = tot_wg = 0
    samples = []
    for _ in range(REPS):
        outs = [torch.zeros((N_WG, OUT_WORDS), dtype=torch.int32, device="cuda") for _ in range(n_canaries)]
        torch.cuda.synchronize()
        fn()
        with torch.cuda.stream(side):
            for out in outs:
                rc = lib.lds_canary_launch(C.c_void_p(out.data_ptr()), N_WG, lds_kb * 1024, spin_us, C.c_void_p(side.cuda_stream))
                assert rc == 0, rc
        fn()
        torch.cuda.synchronize()
        for out in outs:
            o = out.cpu().numpy().view("uint32")
            hit = o[:, 0] > 0
            tot_changed += int(o[:, 0].sum()); tot_wg += int(hit.sum())
            for r in o[hit][:1]:
                samples.append([(int(r[4 + k]), hex(int(r[10 + k]))) for k in range(int(r[3]))])
    print(f"LATE  {name:38s} canary LDS {lds_kb:3d} KB: workgroups hit {tot_wg:5d} of {REPS * N_WG * n_canaries}, words changed {tot_changed:7d}"
          + (f", samples {samples[:3]}" if tot_wg else ""), flush=True)


if os.environ.get("LOADS"):
    # canary 2: VGPR-returning global loads of a workgroup that shares its CU with the GEMM's workgroups
    lib.load_canary_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
    n_words = 4 << 20                                  # 16 MB: L2-resident or not, both
    buf = torch.empty((n_words,), dtype=torch.int32, device="cuda")
    assert lib.load_canary_launch(None, C.c_void_p(buf.data_ptr()), n_words, 1, 0, 0, 0.0, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    torch.cuda.synchronize()

    def run_loads(name, fn, lds_kb=56, spin_us=3000.0):
        bad_wg = bad = passes = 0
        first = None
        for _ in range(REPS):
            out = torch.zeros((N_WG, 8), dtype=torch.int32, device="cuda")
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                rc = lib.load_canary_launch(C.c_void_p(out.data_ptr()), C.c_void_p(buf.data_ptr()), n_words, 0, N_WG, lds_kb * 1024, spin_us, C.c_void_p(side.cuda_stream))
                assert rc == 0, rc
            fn()
            torch.cuda.synchronize()
            o = out.cpu().numpy().view("uint32")
            bad_wg += int((o[:, 0] > 0).sum()); bad += int(o[:, 0].sum()); passes += int(o[:, 4].sum())
            if first is None and (o[:, 0] > 0).any():
                r = o[o[:, 0] > 0][0]
                first = (int(r[1]), hex(int(r[2])), hex(int(r[3])))
        print(f"LOADS {name:44s} LDS {lds_kb:3d} KB: workgroups with a wrong word {bad_wg:5d} of {REPS * N_WG}, wrong words {bad}, 16 x 16-byte load batches checked {passes * 256}"
              + (f", first (word, got, want) {first}" if first else ""), flush=True)

    run_loads("nothing beside it", lambda: None)
    run_loads("op_gemm 640 x 768 x 768 (small tiles, split-K)", lambda: [eng.op_gemm(x640, w768) for _ in range(60)])
    run_loads("op_gemm 2008 x 3072 x 768", lambda: [eng.op_gemm(x2008, w3072) for _ in range(30)])
    run_loads("op_gemm 640 x 768 x 768, canary LDS 24 KB", lambda: [eng.op_gemm(x640, w768) for _ in range(60)], lds_kb=24)
    run_loads("op_attention 5 x 120", lambda: [eng.op_attention(q, q, q, ql) for _ in range(60)])
    sys.exit(0)
if os.environ.get("LATE"):
    run_late("op_gemm 640 x 768 x 768", lambda: [eng.op_gemm(x640, w768) for _ in range(40)])
    run_late("op_gemm 640 x 768 x 768", lambda: [eng.op_gemm(x640, w768) for _ in range(40)], lds_kb=100)
    run_late("op_gemm 2008 x 3072 x 768", lambda: [eng.op_gemm(x2008, w3072) for _ in range(20)])
    run_late("op_gemm 16064 x 768 x 768", lambda: [eng.op_gemm(x16k, w768) for _ in range(4)], lds_kb=120)
    for st in (2, 3):
        eng.lib.gam_tune_sp_stages(st)
        run_late(f"op_gemm 640, stages {st}", lambda: [eng.op_gemm(x640, w768) for _ in range(40)])
    eng.lib.gam_tune_sp_stages(0)
    sys.exit(0)
run("nothing beside it", lambda: None)
run("op_gemm 640 x 768 x 768 (small tiles, split-K)", lambda: [eng.op_gemm(x640, w768) for _ in range(40)])
run("op_gemm 2008 x 3072 x 768", lambda: [eng.op_gemm(x2008, w3072) for _ in range(20)])
run("op_gemm 16064 x 768 x 768 (112 KB tiles)", lambda: [eng.op_gemm(x16k, w768) for _ in range(6)])
run("op_attention 5 x 120", lambda: [eng.op_attention(q, q, q, ql) for _ in range(40)])
for st in (2, 3):
    eng.lib.gam_tune_sp_stages(st)
    run(f"op_gemm 640 x 768 x 768, stages forced {st}", lambda: [eng.op_gemm(x640, w768) for _ in range(40)])
eng.lib.gam_tune_sp_stages(0)
os.environ["X"] = "1"
for kb in (24, 90, 120):
    run("op_gemm 640 x 768 x 768", lambda: [eng.op_gemm(x640, w768) for _ in range(40)], lds_kb=kb)
