#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (…_results.db) into the plain-text per-kernel summary
committed under profiles/ (the equivalent of `rocprofv3 --kernel-trace --stats`)."""
import sqlite3
import sys


def main(db_path, out_path, note=""):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(lds_size), max(grid_x*grid_y*grid_z), max(workgroup_x) from kernels group by name order by 3 desc"))
    total = sum(r[2] for r in rows) or 1
    with open(out_path, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace summary of {db_path}\n# {note}\n")
        f.write("# durations in microseconds (rocpd 'duration' is ns; converted)\n")
        f.write(f"{'kernel':90s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s} {'vgpr':>5s} {'lds':>7s} {'grid':>9s} {'wg':>4s}\n")
        for n, c, tot, avg, mn, mx, vg, lds, grid, wg in rows:
            f.write(f"{n[:90]:90s} {c:6d} {tot/1e3:12.1f} {avg/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*tot/total:6.2f} {vg or 0:5d} {lds or 0:7d} {grid or 0:9d} {wg or 0:4d}\n")
        try:
            cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
            n = list(cur.execute("select count(*) from counters_collection"))[0][0]
            if n:
                f.write("\n# PMC counters: columns " + ",".join(cols) + "\n")
                kcol = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else None)
                ccol = "counter_name" if "counter_name" in cols else None
                if kcol and ccol:
                    for r in cur.execute(f"select {kcol}, {ccol}, count(*), sum(value) from counters_collection group by 1,2 order by 1,2"):
                        f.write(f"{str(r[0])[:90]:90s} {r[1]:28s} n={r[2]:6d} sum={r[3]:.6g} per_dispatch={r[3]/max(1,r[2]):.6g}\n")
        except sqlite3.Error as e:
            f.write(f"# (no counters: {e})\n")
    print(open(out_path).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
