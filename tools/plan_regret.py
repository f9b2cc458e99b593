#!/usr/bin/env python3
"""How good are gam_gemm_sp_plan's picks?  Reads sweeps written by `tools/smallm_sweep.py --calib --stages` (the eight fastest measured
configurations per shape), asks the LIBRARY's planner (gam_plan_sp_ex: host code, no GPU needed) what it would launch for each shape and
prints pick, measured time of the pick, the best measured configuration and the regret.  A pick outside the eight fastest is '?'.
    python tools/plan_regret.py profiles/r06_gemm_sweep_*.txt"""
import ctypes as C
import os
import re
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gigaam_amd import _lib  # noqa: E402

lib = _lib.load_library()
reg, unknown = [], 0
for path in sys.argv[1:]:
    for ln in open(path):
        m = re.match(r"M=\s*(\d+) N=\s*(\d+) K=\s*(\d+)", ln)
        if not m or "best:" not in ln:
            continue
        M, N, K = map(int, m.groups())
        meas = {(int(a), int(b), int(s), 3 if n3 else 2): float(us) for a, b, s, n3, us in re.findall(r"(\d)x(\d)/S(\d)(/n3)? ([\d.]+)", ln.split("best:")[1])}
        mt, nw, s, ns = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        assert lib.gam_plan_sp_ex(M, N, K, 256, C.byref(mt), C.byref(nw), C.byref(s), C.byref(ns)) == 0
        pick = (mt.value, nw.value, s.value, ns.value)
        best = min(meas, key=meas.get)
        v = meas.get(pick)
        if v is None:
            unknown += 1
        else:
            reg.append(v / meas[best])
        mark = "" if v is not None and v <= 1.04 * meas[best] else "   <<<"
        print(f"M={M:6d} N={N:5d} K={K:6d}  pick {pick[0]}x{pick[1]}/S{pick[2]}/n{pick[3]} {('%.1f us' % v) if v else '?':>9s}   best {best[0]}x{best[1]}/S{best[2]}/n{best[3]} {meas[best]:.1f} us{mark}")
print(f"shapes {len(reg) + unknown}: mean regret {statistics.mean(reg):.4f}, worst {max(reg):.4f}; picks outside the eight fastest: {unknown}")
