#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into a per-kernel stats table."""
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
        "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    span = list(cur.execute("select (max(end)-min(start))/1e6 from kernels"))[0][0]
    print(f"# {path}\n# total kernel time {tot:.2f} ms over a span of {span:.2f} ms, {sum(r[1] for r in rows)} dispatches")
    print(f"{'kernel':100s} {'calls':>6s} {'total_ms':>10s} {'pct':>6s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>10s} vgpr agpr lds")
    for r in rows[:top]:
        print(f"{r[0][:100]:100s} {r[1]:6d} {r[2]:10.3f} {100 * r[2] / tot:6.2f} {r[3]:10.1f} {r[4]:9.1f} {r[5]:10.1f} {r[6]} {r[7]} {r[8]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
