#!/usr/bin/env python3
"""Idle gaps and slow outliers on the GPU timeline of a rocprofv3 rocpd database (kernel trace):
    python tools/rocpd_gaps.py <results.db> [min_gap_ms]
Prints every gap >= min_gap_ms between the end of one kernel and the start of the next (with the two kernels), and
the dispatches that took more than 3x their kernel's median duration."""
import sqlite3
import statistics
import sys


def main(db, min_gap_ms=2.0):
    cur = sqlite3.connect(db).cursor()
    cols = [d[1] for d in cur.execute("pragma table_info(kernels)")]
    start = "start" if "start" in cols else "start_timestamp"
    end = "end" if "end" in cols else "end_timestamp"
    rows = list(cur.execute(f"select name, {start}, {end} from kernels order by {start}"))
    t0 = rows[0][1]
    print(f"{len(rows)} dispatches over {(rows[-1][2] - t0) / 1e6:.1f} ms; busy {sum(r[2] - r[1] for r in rows) / 1e6:.1f} ms")
    prev = rows[0]
    for r in rows[1:]:
        gap = (r[1] - prev[2]) / 1e6
        if gap >= min_gap_ms:
            print(f"gap {gap:8.2f} ms at t={(prev[2] - t0) / 1e6:9.1f} ms  after {prev[0][:50]}  before {r[0][:50]}")
        if r[2] > prev[2]:
            prev = r
    by = {}
    for n, s, e in rows:
        by.setdefault(n, []).append((e - s) / 1e3)
    for n, d in by.items():
        med = statistics.median(d)
        out = [x for x in d if x > 3 * med and x > 100]
        if out:
            print(f"outliers {n[:60]}: median {med:.1f} us, {len(out)} dispatches > 3x: max {max(out):.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 2.0)
