#!/usr/bin/env python3
"""Timeline view of a rocprofv3 --kernel-trace CSV: the last busy window of the process with >= 1000 launches (the final repeat of bench.py's timed
region), split into slices; per slice the time at least one kernel was running, the average number of kernels running at once
and the launch count, then the per-kernel totals inside the window.
Usage: python tools/trace_timeline.py KERNEL_TRACE.csv [GAP_MS=5] [SLICES=10]"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    gap = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 5e6
    n_slices = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    ev = []
    with open(path) as f:
        for r in csv.DictReader(f):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Stream_Id"]))
    ev.sort()
    # windows = maximal runs of launches with no idle gap longer than `gap`
    windows, cur_end = [], None
    for s, e, *_ in ev:
        if cur_end is None or s - cur_end > gap:
            windows.append([s, e, 0])
        w = windows[-1]
        w[1] = max(w[1], e)
        w[2] += 1
        cur_end = w[1]
    print("busy windows (ms long, launches):", [(round((w[1] - w[0]) / 1e6, 1), w[2]) for w in windows][-8:])
    big = [w for w in windows if w[2] >= 1000] or windows
    w0, w1, _ = big[-1]           # the last window with real work in it
    inside = [x for x in ev if x[0] >= w0 and x[1] <= w1]
    step = (w1 - w0) / n_slices
    print(f"last window: {(w1 - w0) / 1e6:.2f} ms, {len(inside)} launches on {len(set(x[3] for x in inside))} streams")
    print("slice | busy share | kernels running on average | launches | mean launch us")
    for i in range(n_slices):
        a, b = w0 + i * step, w0 + (i + 1) * step
        pts = []
        n = 0
        durs = []
        for s, e, *_ in inside:
            if e <= a or s >= b:
                continue
            pts.append((max(s, a), 1))
            pts.append((min(e, b), -1))
            if a <= s < b:
                n += 1
                durs.append(e - s)
        pts.sort()
        busy = area = 0
        depth, last = 0, a
        for t, d in pts:
            if depth > 0:
                busy += t - last
            area += depth * (t - last)
            depth += d
            last = t
        print(f"{i:2d} | {busy / step:.2f} | {area / step:.2f} | {n} | {sum(durs) / max(1, len(durs)) / 1e3:.1f}")
    per = defaultdict(lambda: [0, 0])
    for s, e, name, _ in inside:
        per[name][0] += 1
        per[name][1] += e - s
    tot = sum(v[1] for v in per.values())
    print(f"sum of kernel durations in the window: {tot / 1e6:.1f} ms = {tot / (w1 - w0):.2f} x its wall time")
    for name, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:16]:
        print(f"  {name[:60]:60s} {n:6d} launches {t / 1e6:8.2f} ms {100 * t / tot:5.1f}% avg {t / n / 1e3:6.1f} us")


if __name__ == "__main__":
    main()
