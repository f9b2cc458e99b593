#!/usr/bin/env python3
"""Which kernels ran WHILE the RNN-T cluster decode kernel was running?  (rocprofv3 --kernel-trace rocpd database)
    python tools/rocpd_overlap.py <results.db> [pattern=gam_rnnt_cluster_kernel]
For every dispatch of the pattern kernel: its span, the other dispatches whose [start, end] intersects it, the time those cover inside the span
(union), and what ran in the `lead` microseconds before it.  The evidence that the decode of batch n runs beside the encoder of batch n + 1."""
import sqlite3
import sys
from collections import Counter


def main(db, pat="gam_rnnt_cluster_kernel"):
    cur = sqlite3.connect(db).cursor()
    cols = [d[1] for d in cur.execute("pragma table_info(kernels)")]
    start = "start" if "start" in cols else "start_timestamp"
    end = "end" if "end" in cols else "end_timestamp"
    extra = ", queue_id" if "queue_id" in cols else (", stream_id" if "stream_id" in cols else "")
    rows = list(cur.execute(f"select name, {start}, {end}{extra} from kernels order by {start}"))
    t0 = rows[0][1]
    dec = [r for r in rows if pat in r[0]]
    print(f"{len(rows)} dispatches, {len(dec)} of {pat}")
    cover_tot = span_tot = 0.0
    for k, d in enumerate(dec):
        s, e = d[1], d[2]
        inside = [r for r in rows if r is not d and r[2] > s and r[1] < e and pat not in r[0]]
        # union of the overlapping kernels' intervals clipped to the decode's span
        iv = sorted((max(r[1], s), min(r[2], e)) for r in inside)
        cov, cur_s, cur_e = 0, None, None
        for a, b in iv:
            if cur_e is None or a > cur_e:
                if cur_e is not None:
                    cov += cur_e - cur_s
                cur_s, cur_e = a, b
            else:
                cur_e = max(cur_e, b)
        if cur_e is not None:
            cov += cur_e - cur_s
        cover_tot += cov; span_tot += e - s
        if k < 6 or k == len(dec) - 1:
            names = Counter(r[0].split("(")[0].replace("void ", "")[:44] for r in inside)
            qs = sorted({r[3] for r in inside} | {d[3]}) if extra else []
            print(f"decode #{k}: t = {(s - t0) / 1e6:9.3f} ms, span {(e - s) / 1e3:8.1f} us, {len(inside):3d} other dispatches inside it covering {100.0 * cov / (e - s):5.1f} % "
                  f"of the span" + (f"; queues {qs} (decode on {d[3]})" if extra else ""))
            print("          " + ", ".join(f"{n} x{c}" for n, c in names.most_common(6)))
    if span_tot:
        print(f"all {len(dec)} decodes: {100.0 * cover_tot / span_tot:.1f} % of the decode time has another stream's kernels running beside it")


if __name__ == "__main__":
    main(*sys.argv[1:3])
