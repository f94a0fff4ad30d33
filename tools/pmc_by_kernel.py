#!/usr/bin/env python3
"""Per-kernel sums of a rocprofv3 --pmc pass (counter_collection.csv): one row per kernel, one column per counter, plus the
launch count.  Usage: python tools/pmc_by_kernel.py COUNTER_COLLECTION.csv [substring of the kernel names to keep]"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    keep = sys.argv[2] if len(sys.argv) > 2 else ""
    val = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    names = []
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if keep and keep not in k:
                continue
            c = r["Counter_Name"]
            if c not in names:
                names.append(c)
            val[k][c] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    print("kernel | launches | " + " | ".join(names))
    for k in sorted(val, key=lambda k: -val[k].get(names[0], 0)):
        print(f"{k[:50]} | {len(disp[k])} | " + " | ".join(f"{val[k].get(c, 0):.4g}" for c in names))


if __name__ == "__main__":
    main()
