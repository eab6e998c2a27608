#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into a per-kernel stats table (like --stats CSV)."""
import sqlite3
import sys


def main(path, skip_first=0):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    scols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
    q = f"select s.{name_col}, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"
    rows = cur.execute(q).fetchall()
    stats = {}
    for name, st, en in rows:
        name = name.split("(")[0]
        stats.setdefault(name, []).append(en - st)
    total = sum(sum(v) for v in stats.values())
    span = rows[-1][2] - rows[0][1] if rows else 0
    print(f"# {path}: {len(rows)} dispatches, kernel time {total/1e6:.3f} ms, trace span {span/1e6:.3f} ms")
    print(f"{'kernel':60s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for name, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        print(f"{name[:60]:60s} {len(v):7d} {sum(v)/1e6:10.3f} {sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:9.2f} {max(v)/1e3:9.2f} {100*sum(v)/total:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
