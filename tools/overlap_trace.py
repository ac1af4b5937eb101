#!/usr/bin/env python3
"""Concurrency summary of a rocprofv3 --kernel-trace CSV (several batches in flight on several streams):
how much of the wall time has 0 / 1 / 2 / 3+ kernels resident, and per kernel symbol its launches, average duration and the
average number of OTHER kernels running during it.  Usage: overlap_trace.py <..._kernel_trace.csv> [skip_fraction]
The first skip_fraction (default 0.3) of the trace's time span is dropped (engine creation, warm-up)."""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0], r.get("Queue_Id", "")))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    lo = t0 + int((t1 - t0) * skip)
    rows = [r for r in rows if r[0] >= lo]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    ev = []
    for s, e, n, q in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    hist = defaultdict(int)
    level, prev = 0, ev[0][0]
    for t, d in ev:
        hist[min(level, 4)] += t - prev
        prev = t
        level += d
    span = float(t1 - t0)
    print("window %.3f ms, %d kernel launches, queues %s" % (span / 1e6, len(rows), sorted({r[3] for r in rows})))
    print("resident kernels : share of wall time")
    for k in sorted(hist):
        print("  %s%d : %.3f" % (">=" if k == 4 else "  ", k, hist[k] / span))
    busy = sum(e - s for s, e, _, _ in rows)
    print("sum of kernel durations / wall = %.3f" % (busy / span))
    # per symbol: average duration, average overlap with other kernels (time-weighted count of others)
    by = defaultdict(list)
    starts = [(s, e) for s, e, _, _ in rows]
    for i, (s, e, n, q) in enumerate(rows):
        others = 0
        for j in range(max(0, i - 40), min(len(rows), i + 40)):
            if j == i:
                continue
            s2, e2 = starts[j]
            ov = min(e, e2) - max(s, s2)
            if ov > 0:
                others += ov
        by[n].append((e - s, others / max(e - s, 1)))
    print("%-64s %6s %9s %9s %7s" % ("kernel", "calls", "avg us", "total ms", "others"))
    for n, v in sorted(by.items(), key=lambda kv: -sum(d for d, _ in kv[1])):
        tot = sum(d for d, _ in v)
        print("%-64s %6d %9.1f %9.3f %7.2f" % (n[:64], len(v), tot / len(v) / 1e3, tot / 1e6, sum(o * d for d, o in v) / tot))


if __name__ == "__main__":
    main()
