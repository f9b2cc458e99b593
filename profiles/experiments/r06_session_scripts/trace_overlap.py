#!/usr/bin/env python3
"""How much of a rocprofv3 kernel trace (rocpd .db) had two or more kernels in flight?  (r06: do the launch chains of two streams overlap?)
    python tools/trace_overlap.py trace.db"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cols = [d[1] for d in db.execute("pragma table_info(kernels)")]
    qcol = next((c for c in ("queue_id", "queue", "stream_id", "stream") if c in cols), None)
    rows = list(db.execute(f"select start, end, {qcol or '0'}, name from kernels order by start"))
    print("kernels:", len(rows), " queues:", sorted({r[2] for r in rows}))
    if "stream_id" in cols and "queue_id" in cols:
        print("(stream_id, queue_id): kernels ->", {k: v for k, v in db.execute("select stream_id || '/' || queue_id, count(*) from kernels group by 1")})
    ev = []
    for s, e, _, _ in rows:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    depth, last, busy1, busy2 = 0, ev[0][0], 0, 0
    for t, d in ev:
        if depth >= 1: busy1 += t - last
        if depth >= 2: busy2 += t - last
        depth += d; last = t
    tot = sum(e - s for s, e, _, _ in rows)
    print(f"sum of kernel durations {tot/1e6:.3f} ms; time with >= 1 kernel in flight {busy1/1e6:.3f} ms; with >= 2 in flight {busy2/1e6:.3f} ms "
          f"({100.0*busy2/max(1,busy1):.1f} %)")
    # per queue: gap between consecutive kernels
    byq = {}
    for s, e, q, n in rows:
        byq.setdefault(q, []).append((s, e, n))
    for q, v in sorted(byq.items()):
        gaps = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1) if v[i + 1][0] - v[i][1] < 50000]
        if gaps:
            gaps.sort()
            print(f"queue {q}: {len(v)} kernels, gap between consecutive kernels (under 50 us): median {gaps[len(gaps)//2]/1e3:.2f} us, mean {sum(gaps)/len(gaps)/1e3:.2f} us, "
                  f"p90 {gaps[int(0.9*len(gaps))]/1e3:.2f} us, sum {sum(gaps)/1e6:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1])
