#!/usr/bin/env python3
"""Where a stream idles in a rocprofv3 --kernel-trace CSV of bench.py with several batches in flight: per queue the gaps between
consecutive kernels, split into gaps INSIDE a batch (dependent launches of one submit) and gaps BETWEEN batches (from the last
kernel of a batch to the first kernel of the next one on that stream: collect + submit on the host).  The first and last `trim`
fraction of the time span is dropped (warm-up, drain).  Usage: stream_gaps.py <..._kernel_trace.csv> [trim=0.15]"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    trim = float(sys.argv[2]) if len(sys.argv) > 2 else 0.15
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "")))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    lo, hi = t0 + int((t1 - t0) * trim), t1 - int((t1 - t0) * trim)
    byq = defaultdict(list)
    for r in rows:
        if r[0] >= lo and r[1] <= hi:
            byq[r[3]].append(r)
    span = (hi - lo) / 1e6
    print("window %.2f ms" % span)
    tot_intra = tot_inter = 0.0
    for q, v in sorted(byq.items()):
        if len(v) < 50:
            continue
        intra, inter, busy, n_inter = 0.0, 0.0, 0.0, 0
        biggest = []
        for a, b in zip(v[:-1], v[1:]):
            gap = (b[0] - a[1]) / 1e3
            busy += (a[1] - a[0]) / 1e3
            # the last kernel of a batch is the SparseTensor scatter (or the scan when nnz = 0); the first is the table conv / a fill
            if "scatter_kernel" in a[2] or ("scan_kernel" in a[2] and "scatter" not in b[2]):
                inter += max(gap, 0.0)
                n_inter += 1
            else:
                intra += max(gap, 0.0)
                biggest.append((gap, a[2][-40:], b[2][-40:]))
        biggest.sort(reverse=True)
        q_span = (v[-1][1] - v[0][0]) / 1e3
        print("queue %s: %d kernels over %.1f ms: busy %.1f %%, gaps inside batches %.2f %% (%.1f us per kernel), between batches %.2f %% "
              "(%d batches, %.0f us each)" % (q, len(v), q_span / 1e3, 100 * busy / q_span, 100 * intra / q_span, intra / len(v),
                                              100 * inter / q_span, n_inter, inter / max(n_inter, 1)))
        for g, a, b in biggest[:4]:
            print("      %.1f us  after ..%s  before ..%s" % (g, a, b))
        tot_intra += intra / q_span
        tot_inter += inter / q_span
    print("mean over queues: inside batches %.2f %%, between batches %.2f %%" % (100 * tot_intra / max(1, len([1 for v in byq.values() if len(v) >= 50])),
                                                                              100 * tot_inter / max(1, len([1 for v in byq.values() if len(v) >= 50]))))


if __name__ == "__main__":
    main()
