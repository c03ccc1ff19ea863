#!/usr/bin/env python
"""Summarises a rocprofv3 rocpd database (kernel trace) into a per-kernel table:  python tools/rocpd_summary.py x.db"""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = [f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}"]
    for n, c, s, a, mn, mx in rows:
        short = n.replace("(anonymous namespace)::", "").replace("void ", "")
        short = short.split("(")[0][-70:]
        lines.append(f"{short:70s} {c:6d} {s / 1e6:10.3f} {a / 1e3:10.1f} {mn / 1e3:10.1f} {mx / 1e3:10.1f} {100 * s / total:6.2f}")
    lines.append(f"{'TOTAL':70s} {sum(r[1] for r in rows):6d} {total / 1e6:10.3f}")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
