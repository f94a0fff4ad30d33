"""HBM-side traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output).
Usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> ["title"] [gemm.json]
With a fourth argument also writes the per-launch average over all k_gemm* launches as JSON (bench.py reports it as
roofline.traffic).
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
    if len(sys.argv) > 4:
        import json
        nf = sum(n for (name, _), (n, v) in fetch.items() if "k_gemm" in name)
        fv = sum(v for (name, _), (n, v) in fetch.items() if "k_gemm" in name)
        nw = sum(n for (name, _), (n, v) in write.items() if "k_gemm" in name)
        wv = sum(v for (name, _), (n, v) in write.items() if "k_gemm" in name)
        out = {"source": sys.argv[3] if len(sys.argv) > 3 else "", "launches": nf, "fetch_size_kb_sum": fv, "write_size_kb_sum": wv,
               "bytes_per_launch": 2 * fv * 1024 / max(nf, 1) + wv * 1024 / max(nw, 1),
               "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads, MI355X_MICROARCH.md HBM section); counters are in KB; "
                             "Infinity-Cache hits are counted"}
        with open(sys.argv[4], "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
