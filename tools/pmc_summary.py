"""HBM-side traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output).
Usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> ["title"]
Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): both
counters are in KB; FETCH_SIZE reports half the bytes of a wide coalesced stream (the x2 column applies that)."""
import csv
import sys
from collections import defaultdict


def load(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            wgs = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
            a = acc[(name, wgs)]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    print(sys.argv[3] if len(sys.argv) > 3 else "")
    print("kernel | workgroups | launches | FETCH_SIZE KB/launch | FETCH x2 KB/launch | WRITE_SIZE KB/launch | total x2-fetch+write MB")
    rows = []
    for k, (n, v) in fetch.items():
        wn, wv = write.get(k, [0, 0.0])
        rows.append((2 * v + wv, k, n, v / n, wv / max(1, wn)))
    for tot, (name, wgs), n, f, w in sorted(rows, reverse=True)[:24]:
        print(f"{name} | {wgs} | {n} | {f:.0f} | {2 * f:.0f} | {w:.0f} | {tot / 1e3:.0f}")


if __name__ == "__main__":
    main()
