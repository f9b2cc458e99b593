#!/usr/bin/env python3
"""Fit the constants of gam_gemm_sp_model (gigaam_amd/csrc/gam_gemm_sp.h) to a sweep of tools/smallm_sweep.py --calib.

    python tools/fit_sp_model.py [gpurun_out/smallm_sweep.json]

Model (us): rounds x (k-tiles-per-slice x t_kt + t_fix) [+ reduce pass when S > 1], per tile class
  nw 4: 8 waves, (64 mt) x 256     t_kt = A4 mt + B4,                      t_fix = F4 | F4S
  nw 2: 4 waves, (64 mt) x 128     t_kt = (A2 mt + B2) (x SH2 when two workgroups share a CU), t_fix = F2 | F2S (x SHF2)
  (a third class -- 128 wide with EIGHT waves, a wave owning 32 mt x 32 -- was built and measured in r03: within 1 % of the
   4-wave class at equal tile shape, 4 % better only at 126 rows; dropped.  Configurations named "MTx3" in a sweep file are ignored.)
  reduce: R0 + (S + 2) M N 4 B / RBW
Random-restart coordinate search on the mean squared log error; prints the #defines and the plan's regret (time of the
configuration the fitted model picks / best measured) per shape."""
import json
import math
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["A4", "B4", "F4", "F4S", "A2", "B2", "F2", "F2S", "SH2", "SHF2", "R0", "RBW"]
INIT = [0.335, 0.425, 13.2, 6.2, 0.256, 0.142, 14.1, 4.9, 1.82, 0.73, 7.1, 3.87]


def cdiv(a, b):
    return (a + b - 1) // b


def model(p, M, N, K, t, w, S, ncu=256):
    P = dict(zip(NAMES, p))
    bn = 256 if w == 4 else 128
    wgs = cdiv(M, 64 * t) * cdiv(N, bn) * S
    nkt = K // 32 // S
    if w == 4:
        tk, tf, rounds = P["A4"] * t + P["B4"], (P["F4S"] if S > 1 else P["F4"]), cdiv(wgs, ncu)
    else:
        c = "2"
        shared = wgs > ncu
        tk = (P["A" + c] * t + P["B" + c]) * (P["SH" + c] if shared else 1.0)
        tf = (P["F" + c + "S"] if S > 1 else P["F" + c]) * (P["SHF" + c] if shared else 1.0)
        rounds = cdiv(wgs, (2 if shared else 1) * ncu)
    us = rounds * (nkt * tk + tf)
    if S > 1:
        us += P["R0"] + (S + 2) * M * N * 4.0 / (P["RBW"] * 1e6)
    return us


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "smallm_sweep.json")
    rows = json.load(open(path))
    data = []
    for r in rows:
        for c, t in r.get("configs", {}).items():
            mt, rest = c.split("x")
            nw, s = rest.split("/S")
            if int(nw) in (2, 4):
                data.append((r["M"], r["N"], r["K"], int(mt), int(nw), int(s), t))

    def loss(p):
        return sum(math.log(model(p, *d[:6]) / d[6]) ** 2 for d in data) / len(data)

    random.seed(0)
    p, best = list(INIT), loss(INIT)
    for _ in range(40000):
        q = list(p)
        i = random.randrange(len(p))
        q[i] *= math.exp(random.gauss(0, 0.08))
        l = loss(q)
        if l < best:
            best, p = l, q
    print(f"{len(data)} points, r.m.s. log error {math.sqrt(best):.3f}")
    for n, v in zip(NAMES, p):
        print(f"#define GAM_SPM_{n} {v:.3g}")
    tot = n_ = 0
    for r in rows:
        cf = r.get("configs", {})
        if not cf:
            continue
        pred = {}
        for c in cf:
            mt, rest = c.split("x")
            nw, s = rest.split("/S")
            mt, nw, s = int(mt), int(nw), int(s)
            if nw not in (2, 4):
                continue
            tiles = cdiv(r["M"], 64 * mt) * cdiv(r["N"], 256 if nw == 4 else 128)
            if s > 1 and tiles * 2 > 256:
                continue
            pred[c] = model(p, r["M"], r["N"], r["K"], mt, nw, s)
        ch, bt = min(pred, key=pred.get), min(cf.values())
        print(f"M={r['M']:6d} N={r['N']:5d} K={r['K']:6d} plan {ch:8s} {cf[ch]:7.1f} us | best {min(cf, key=cf.get):8s} {bt:7.1f} | regret {cf[ch] / bt:.2f}")
        tot += cf[ch] / bt
        n_ += 1
    print(f"mean regret {tot / max(1, n_):.3f}")


if __name__ == "__main__":
    main()
