#!/usr/bin/env python3
"""One line per kernel from hipcc's -Rpass-analysis=kernel-resource-usage remarks (stdin or a file): name, VGPRs, AGPRs,
scratch bytes per lane, occupancy, LDS.   hipcc ... -Rpass-analysis=kernel-resource-usage 2>&1 | python tools/kres.py [filter]"""
import re
import subprocess
import sys

txt = sys.stdin.read()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None
rows = []
for ln in txt.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]): (\S+)", ln)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" ")[0] + ("s" if k.endswith("Spill") else "")] = v
names = [r["name"] for r in rows]
try:
    dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + names, capture_output=True, text=True).stdout.splitlines()
except Exception:
    dem = names
for r, d in zip(rows, dem):
    d = re.sub(r"\(.*", "", d).replace("void ", "")
    if flt and flt not in d:
        continue
    print(f"{d:55s} vgpr {r.get('VGPRs','?'):>4s} agpr {r.get('AGPRs','?'):>4s} scratch {r.get('ScratchSize','?'):>4s} occ {r.get('Occupancy','?'):>2s} lds {r.get('LDS','?'):>6s}")
