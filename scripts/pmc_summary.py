#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter CSVs per kernel name: prints one line per (kernel, counter) with
the total over dispatches and the mean per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    tot = defaultdict(lambda: defaultdict(float))
    ndisp = defaultdict(lambda: defaultdict(set))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = row.get("Kernel_Name", "?").split("(")[0]
                c = row.get("Counter_Name", "?")
                tot[k][c] += float(row.get("Counter_Value", 0) or 0)
                ndisp[k][c].add(row.get("Dispatch_Id"))
    for k in sorted(tot):
        print(f"== {k}")
        for c in sorted(tot[k]):
            n = max(len(ndisp[k][c]), 1)
            print(f"   {c:36s} total {tot[k][c]:16.0f}  per-dispatch {tot[k][c] / n:16.1f}  (dispatches {n})")


if __name__ == "__main__":
    main()
