#!/usr/bin/env python3
"""The HBM-bound kernels of the headline step against the roofline (VERDICT r3 "next" #5c): algorithmic bytes per launch
(unique reads + writes, from the shapes) / rocprofv3 kernel-trace average duration, against 8 TB/s (spec) and the 6.29 TB/s
a float4 copy reaches on this part (MI355X_MICROARCH.md).  FETCH / WRITE columns come from the PMC passes when given.

    python tools/hbm_kernels.py TRACE_SUMMARY [FETCH_SUMMARY WRITE_SUMMARY] > profiles/r04_hbm_kernels.txt

TRACE_SUMMARY etc. are the files tools/rocpd_summary.py writes (profiles/r0N_trace_summary.txt, r0N_f16x3_fetch.txt ...)."""
import re
import sys

B, SEC, D, H = 32, 20.0, 768, 16
T = 2001                      # log-mel frames of a 20 s utterance (hop 160, center=True)
T1, TV = 1001, 501
TA = 502                      # row stride per utterance (gam_api.hip encode_impl)
N = B * TA                    # token rows
FP, F2 = 33, 16               # stage-1 feature bins + border, stage-2 bins
NF, LDS = 201, 404            # DFT bins, spectrum row pitch (floats)
TFA = 2003                    # frontend row stride per utterance

MB = 1e6
# kernel-name regex -> (label, algorithmic bytes, note)
ROWS = [
    (r"gam_layernorm_kernel<0, false", "gam_layernorm_kernel<0>", N * D * 4 * 2 + N * 4, "reads x (fp32), writes y (sp32, row-scaled) + rs"),
    (r"gam_layernorm_kernel<1, false", "gam_layernorm_kernel<1> (RoPE)", N * D * 4 * 3 + N * 4, "y and rope(y)"),
    (r"gam_layernorm_kernel<2, false", "gam_layernorm_kernel<2> (fused)", N * D * 4 * 3 + N * 4, "norm_out -> x', next norm_feed_forward1 -> y"),
    (r"gam_convmod_bn_kernel<31>", "gam_convmod_bn_kernel<31>", N * D * 4 * 3, "reads [N,1536] GLU input, writes [N,768] (sp32); 23 % halo re-reads stay in L2"),
    (r"gam_conv2d1_kernel", "gam_conv2d1_kernel", B * 2 * TA * FP * D * 4 + B * 64 * T * 4, "writes the zero-bordered channels-last image"),
    (r"gam_powmel_kernel", "gam_powmel_kernel", B * T * (2 * NF + 64) * 4, "reads the spectrum, writes log-mel; filters in LDS"),
    (r"gam_pad_wav_kernel", "gam_pad_wav_kernel", B * (320000 + TFA * 160) * 4, "reflect-padded copy of the waveform"),
    (r"gam_transpose_kernel", "gam_transpose_kernel", B * TV * D * 4 * 2, "token-major <-> channel-first at the C-ABI boundary"),
]


def parse(path):
    """-> {kernel name: (calls, avg_us)} and {kernel name: per-dispatch PMC value} where present"""
    trace, pmc = {}, {}
    for ln in open(path):
        if ln.startswith("#") or ln.startswith("kernel"):
            continue
        m = re.match(r"^(.{90}) +(\d+) +([\d.]+) +([\d.]+) ", ln)
        if m:
            trace[m.group(1).strip()] = (int(m.group(2)), float(m.group(4)))
        m = re.match(r"^(.{90}) (\S+) +n= *(\d+) sum=(\S+) per_dispatch=(\S+)", ln)
        if m:
            pmc[(m.group(1).strip(), m.group(2))] = float(m.group(5))
    return trace, pmc


def main():
    trace, _ = parse(sys.argv[1])
    fetch = parse(sys.argv[2])[1] if len(sys.argv) > 2 else {}
    write = parse(sys.argv[3])[1] if len(sys.argv) > 3 else {}
    print("# HBM-bound kernels of the headline step (v2_ctc, 32 x 20 s, N = %d token rows, d = 768)." % N)
    print("# time    = rocprofv3 --kernel-trace average (%s)" % sys.argv[1])
    print("# alg     = algorithmic bytes per launch (unique reads + writes); achieved = alg / time, against 8 TB/s (spec) and 6.29 TB/s")
    print("#           (what a float4 copy reaches on this part, MI355X_MICROARCH.md)")
    if fetch:
        print("# FETCH / WRITE = rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE per dispatch, separate passes, KiB -> MB; FETCH doubled per the guide's")
        print("#           gfx950 correction.  L2-miss-side counters, Infinity-Cache hits included.")
    print("#")
    print(f"# {'kernel':34s} {'calls':>6s} {'time us':>8s} {'alg MB':>8s} {'TB/s':>6s} {'of 8':>5s} {'of 6.29':>7s} {'FETCHx2 MB':>10s} {'WRITE MB':>9s}  note")
    for rx, label, nbytes, note in ROWS:
        hit = [(k, v) for k, v in trace.items() if re.search(rx, k)]
        if not hit:
            continue
        k, (calls, avg) = hit[0]
        tbs = nbytes / (avg * 1e-6) / 1e12
        f = next((v for (kk, c), v in fetch.items() if kk == k and "FETCH" in c), None)
        w = next((v for (kk, c), v in write.items() if kk == k and "WRITE" in c), None)
        fs = f"{f * 1024 * 2 / MB:10.1f}" if f is not None else f"{'-':>10s}"
        ws = f"{w * 1024 / MB:9.1f}" if w is not None else f"{'-':>9s}"
        print(f"  {label:34s} {calls:6d} {avg:8.1f} {nbytes / MB:8.1f} {tbs:6.2f} {tbs / 8.0:5.2f} {tbs / 6.29:7.2f} {fs} {ws}  {note}")


if __name__ == "__main__":
    main()
