#!/usr/bin/env python3
"""Where do the __amd_rocclr_copyBuffer dispatches of a step come from?  

    rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d DIR -o b -- python bench.py ...
    python tools/copybuffer_origin.py DIR [pattern]

Joins the kernel trace with the HIP API trace by correlation id and prints, for every dispatch whose kernel name matches
``pattern`` (default: copyBuffer), the HIP call that enqueued it and the kernels dispatched right before / after it."""
import collections
import csv
import glob
import os
import sys


def load(d, suffix):
    fs = glob.glob(os.path.join(d, "**", f"*{suffix}"), recursive=True)
    if not fs:
        return []
    with open(fs[0], newline="") as f:
        return list(csv.DictReader(f))


def main():
    d = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else "copyBuffer"
    kern = load(d, "kernel_trace.csv")
    api = load(d, "hip_api_trace.csv")
    if not kern:
        print("no kernel trace csv under", d)
        return
    kcol = "Kernel_Name" if "Kernel_Name" in kern[0] else "kernel_name"
    ccol = next((c for c in kern[0] if c.lower() == "correlation_id"), None)
    scol = next((c for c in kern[0] if c.lower() == "start_timestamp"), None)
    by_corr = {}
    if api:
        fcol = next((c for c in api[0] if c.lower() == "function"), None)
        acc = next((c for c in api[0] if c.lower() == "correlation_id"), None)
        for r in api:
            by_corr[r[acc]] = r[fcol]
    kern.sort(key=lambda r: int(r[scol]))
    short = lambda n: n.split("(")[0].replace("void ", "")[:60]  # noqa: E731
    origin, before, after = collections.Counter(), collections.Counter(), collections.Counter()
    n = 0
    for i, r in enumerate(kern):
        if pat not in r[kcol]:
            continue
        n += 1
        origin[by_corr.get(r[ccol], "?") if ccol else "?"] += 1
        before[short(kern[i - 1][kcol]) if i else "-"] += 1
        after[short(kern[i + 1][kcol]) if i + 1 < len(kern) else "-"] += 1
    print(f"{n} dispatches matching '{pat}' of {len(kern)} kernel dispatches; HIP API rows: {len(api)}")
    for title, c in (("enqueued by", origin), ("dispatched right after", before), ("followed by", after)):
        print(f"  {title}:")
        for k, v in c.most_common(12):
            print(f"    {v:6d}  {k}")
    if api:
        fc = collections.Counter(r[fcol] for r in api)
        print("  HIP API calls in the whole run (top 15):")
        for k, v in fc.most_common(15):
            print(f"    {v:6d}  {k}")


if __name__ == "__main__":
    main()
