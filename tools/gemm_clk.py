#!/usr/bin/env python3
"""Condense the lines a CHIRON_SENS=1024 build of gemm.hip prints (cycle counters of wave 0 of four workgroups per launch):
per kernel shape the mean cycles per tile, in the epilogue, and in the barrier in front of each chunk.  usage: gemm_clk.py <log>..."""
import collections
import re
import sys

for f in sys.argv[1:]:
    d = collections.defaultdict(list)
    for l in open(f):
        m = re.search(r"zout (\d) K (\d+) block \d+: tiles (\d+), cycles per tile: total (\d+), epilogue (\d+), barriers ([\d ]+)", l)
        if m:
            d[(m.group(1), m.group(2))].append([int(m.group(3)), int(m.group(4)), int(m.group(5))] + [int(x) for x in m.group(6).split()])
    print(f)
    for k, v in sorted(d.items()):
        n = len(v)
        mean = [sum(x[i] for x in v) / n for i in range(len(v[0]))]
        print("  zout %s K %s: %d samples, tiles %.1f, per tile: total %.0f epilogue %.0f barriers %s (sum %.0f)"
              % (k[0], k[1], n, mean[0], mean[1], mean[2], " ".join("%.0f" % x for x in mean[3:]), sum(mean[3:])))
