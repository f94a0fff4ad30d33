#!/usr/bin/env python3
"""The N largest launches (by the first counter) of the kernels matching a substring in a rocprofv3 --pmc pass, all counters of
each, and the kernel-trace duration when the matching kernel_trace.csv is given.
Usage: python tools/pmc_top_dispatches.py COUNTER_COLLECTION.csv SUBSTRING [N=6] [KERNEL_TRACE.csv]"""
import csv
import sys
from collections import defaultdict


def main():
    path, keep = sys.argv[1], sys.argv[2]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    dur = {}
    if len(sys.argv) > 4:
        with open(sys.argv[4]) as f:
            for r in csv.DictReader(f):
                dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    val = defaultdict(dict)
    name = {}
    names = []
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if keep not in k:
                continue
            c = r["Counter_Name"]
            if c not in names:
                names.append(c)
            val[r["Dispatch_Id"]][c] = val[r["Dispatch_Id"]].get(c, 0.0) + float(r["Counter_Value"])
            name[r["Dispatch_Id"]] = k
    key = "SQ_WAVE_CYCLES" if "SQ_WAVE_CYCLES" in names else names[0]
    top = sorted(val, key=lambda d: -val[d].get(key, 0))[:n]
    print("dispatch | kernel | us | " + " | ".join(names))
    for d in top:
        print(f"{d} | {name[d][:30]} | {dur.get(d, 0):.1f} | " + " | ".join(f"{val[d].get(c, 0):.4g}" for c in names))


if __name__ == "__main__":
    main()
