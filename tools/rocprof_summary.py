#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) as a per-kernel table:
calls, total/avg/min/max duration.  Usage: rocprof_summary.py results.db [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, n, tot, avg, mn, mx in rows:
        short = name.split("(")[0]
        lines.append("| %s | %d | %.3f | %.1f | %.1f | %.1f | %.1f |" % (short, n, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    text = "\n".join(lines)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
