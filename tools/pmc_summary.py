#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel: mean of each counter per dispatch."""
import csv
import glob
import sys
from collections import defaultdict


def main():
    out = defaultdict(lambda: defaultdict(list))
    for path in sorted(glob.glob(sys.argv[1] + "/*counter_collection.csv")):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0]
            out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(out):
        print(k)
        for c in sorted(out[k]):
            v = out[k][c]
            print("   %-28s n=%-4d mean=%.6g" % (c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
