#!/usr/bin/env python3
"""HBM-side traffic of the GEMM kernel family from two rocprofv3 PMC passes (rocpd .db):
    python tools/pmc_traffic.py <fetch.db> <write.db> <out.json>
FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts 64 B per 128 B request of
a wide (16 B/lane) streaming read, so it is doubled (MI355X_MICROARCH.md §HBM); WRITE_SIZE is
taken as is (it matched the algorithmic C bytes exactly in calibration: 197.4 MB for FFN-up)."""
import json
import sqlite3
import sys


def family_sum(db_path, counter, pattern="gam_gemm"):
    cur = sqlite3.connect(db_path).cursor()
    cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    rows = list(cur.execute(f"select {kcol}, count(*), sum(value) from counters_collection where counter_name=? group by 1", (counter,)))
    n = sum(r[1] for r in rows if pattern in r[0])
    tot = sum(r[2] for r in rows if pattern in r[0])
    per = {r[0]: {"launches": r[1], "kib_per_launch": r[2] / r[1]} for r in rows if pattern in r[0]}
    return n, tot, per


def main(fetch_db, write_db, out):
    nf, fetch_kib, pf = family_sum(fetch_db, "FETCH_SIZE")
    nw, write_kib, pw = family_sum(write_db, "WRITE_SIZE")
    res = {
        "kernel_family": "gam_gemm_* (plain + implicit-GEMM conv)",
        "launches_fetch_pass": nf, "launches_write_pass": nw,
        "fetch_bytes_per_launch_raw": fetch_kib * 1024 / max(1, nf),
        "fetch_bytes_per_launch_corrected_x2": 2 * fetch_kib * 1024 / max(1, nf),
        "write_bytes_per_launch": write_kib * 1024 / max(1, nw),
        "per_kernel_fetch": pf, "per_kernel_write": pw,
        "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of `python bench.py --steps 1 --warmup 1 --cpu-utts 0 --no-profile`; FETCH_SIZE x2 per the gfx950 correction; L2-miss side (Infinity-Cache hits included)",
    }
    res["traffic_bytes_per_launch"] = res["fetch_bytes_per_launch_corrected_x2"] + res["write_bytes_per_launch"]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if not k.startswith("per_kernel")}, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
