#!/usr/bin/env python3
"""Per-shape summary of the GEMM launches of bench.py's roofline pass, from rocprofv3 output of the SAME command.

bench.py measures the dominant kernel family (k_gemm24 / k_gemm2 / k_gemm3: fp32 MFMA GEMMs) in a further run of the timed
region's own call on sessions that launch eagerly — the same pools and streams as the timed region.  With --no-cpu-baseline
(and --no-sub-records for c2) that run is the last GPU work of the process, so in rocprofv3's kernel trace the pass is the LAST
`roofline.launches` GEMM launches (the same number before them is its warm run: same shapes, checked here).  This tool prints,
for those launches, launches / total / average duration per (kernel, grid), the FLOP-weighted figure bench.py calls
`roofline.achieved` recomputed from the trace, and — with the two --pmc passes — HBM-side bytes per launch
(2 x FETCH_SIZE + WRITE_SIZE, both in KB; FETCH_SIZE doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950).

    python tools/roofline_from_trace.py TRACE.csv BENCH.jsonl [--fetch FETCH_cc.csv FETCH_trace.csv] [--write WRITE_cc.csv WRITE_trace.csv]
                                        [--pmc-json OUT.json]
"""
from __future__ import annotations

import argparse
import csv
import json
import sys
from collections import defaultdict


def short(name: str) -> str:
    return name.split("(")[0].replace("void ", "")


def load_trace(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append({"stream": r["Stream_Id"], "dispatch": r["Dispatch_Id"], "name": short(r["Kernel_Name"]),
                         "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "start": int(r["Start_Timestamp"]),
                         "grid": tuple(int(r[f"Grid_Size_{a}"]) // max(1, int(r[f"Workgroup_Size_{a}"])) for a in "XYZ")})
    return rows


def is_gemm(name: str) -> bool:
    return "k_gemm" in name


def pick_pass(rows, want_launches: int):
    """The last `want_launches` GEMM launches of the process (by start time) and the same number before them (the warm run)."""
    gem = sorted((r for r in rows if is_gemm(r["name"])), key=lambda r: r["start"])
    return gem[-want_launches:], gem[-2 * want_launches:-want_launches]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("bench_json")
    ap.add_argument("--fetch", nargs=2, metavar=("COUNTERS", "TRACE"))
    ap.add_argument("--write", nargs=2, metavar=("COUNTERS", "TRACE"))
    ap.add_argument("--pmc-json")
    ap.add_argument("--command", default="")
    a = ap.parse_args()
    line = json.loads([l for l in open(a.bench_json).read().strip().split("\n") if l.startswith("{")][-1])
    roof = line["roofline"]
    rows = load_trace(a.trace)
    gem, warm = pick_pass(rows, int(roof["launches"]))
    print(f"command: {a.command}")
    print(f"bench.py line of the profiled run (slower than an unprofiled run: the profiler serialises kernels): value {line['value']:.1f} "
          f"{line['unit']}; roofline.launches {roof['launches']}, avg_launch_us {roof['avg_launch_us']:.2f} (HIP event pairs net of the "
          f"{roof.get('event_pair_overhead_us', 0):.2f} us an empty pair measures; raw pairs "
          f"{roof.get('avg_launch_us_raw_event_pairs', float('nan')):.2f} us), achieved {roof['achieved']:.1f} {roof['unit']} = "
          f"{roof['frac']:.3f} of {roof['peak']}")
    def shape_counts(rs):
        c = defaultdict(int)
        for r in rs:
            c[(r["name"], r["grid"])] += 1
        return c
    same = shape_counts(gem) == shape_counts(warm)
    print(f"-> roofline pass = the last {len(gem)} GEMM launches of the process on streams {sorted(set(r['stream'] for r in gem))}; "
          f"the {len(warm)} launches before them (its warm run) have {'the same' if same else 'DIFFERENT'} (kernel, grid) counts")
    t_lo, t_hi = min(r["start"] for r in gem), max(r["start"] + r["dur"] for r in gem)
    sel = [r for r in rows if t_lo <= r["start"] <= t_hi]
    tot = sum(r["dur"] for r in gem)
    avg_us = tot / max(1, len(gem)) / 1e3
    flops = roof["flops_per_launch"] * roof["launches"]
    ach = flops / (tot * 1e-9) / 1e12 if tot else 0.0
    print(f"GEMM launches of the pass: {len(gem)}, total {tot / 1e6:.2f} ms, average {avg_us:.2f} us per launch (rocprofv3 kernel trace)")
    print(f"algorithmic GEMM work of the pass (bench.py counters): {flops / 1e12:.3f} TFLOP -> {ach:.1f} TFLOP/s = {ach / roof['peak']:.3f} "
          f"of the {roof['peak']} TFLOP/s fp32 MFMA peak by the trace (bench.py, same run: {roof['frac']:.3f} net of the event-pair "
          f"overhead, {roof.get('frac_raw_event_pairs', float('nan')):.3f} by raw event pairs)")
    allk = defaultdict(lambda: [0, 0])
    for r in sel:
        allk[r["name"]][0] += 1
        allk[r["name"]][1] += r["dur"]
    tall = sum(v[1] for v in allk.values())
    print("\nall kernels launched during the pass: kernel | launches | total ms | share | avg us")
    for k, (n, t) in sorted(allk.items(), key=lambda kv: -kv[1][1])[:16]:
        print(f"{k} | {n} | {t / 1e6:.2f} | {100 * t / tall:.1f}% | {t / n / 1e3:.2f}")
    shapes = defaultdict(lambda: [0, 0])
    for r in gem:
        shapes[(r["name"], r["grid"])][0] += 1
        shapes[(r["name"], r["grid"])][1] += r["dur"]
    print("\nGEMM launches of the pass per (kernel, grid in workgroups x,y,z = column tiles, row-capacity tiles, K slices): launches | total ms | avg us")
    for (k, g), (n, t) in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:24]:
        print(f"{k} | {g[0]},{g[1]},{g[2]} | {n} | {t / 1e6:.2f} | {t / n / 1e3:.2f}")

    if a.fetch and a.write:
        def counter_sum(cc_path, tr_path, counter, want):
            tr = load_trace(tr_path)
            ids = {r["dispatch"] for r in pick_pass(tr, want)[0]}
            n, v = 0, 0.0
            with open(cc_path) as f:
                for r in csv.DictReader(f):
                    if r["Counter_Name"] == counter and r["Dispatch_Id"] in ids:
                        n += 1
                        v += float(r["Counter_Value"])
            return n, v
        nf, fv = counter_sum(a.fetch[0], a.fetch[1], "FETCH_SIZE", int(roof["launches"]))
        nw, wv = counter_sum(a.write[0], a.write[1], "WRITE_SIZE", int(roof["launches"]))
        bpl = 2 * fv * 1024 / max(1, nf) + wv * 1024 / max(1, nw)
        print(f"\nPMC (separate passes of the same command, GEMM launches of the roofline pass): FETCH_SIZE {fv / 1e3:.1f} MB over {nf} "
              f"launches, WRITE_SIZE {wv / 1e3:.1f} MB over {nw} launches -> 2 x FETCH + WRITE = {bpl / 1e6:.2f} MB per launch "
              f"(algorithmic operand bytes per launch by bench.py: {roof['algorithmic_bytes_per_launch'] / 1e6:.2f} MB)")
        if a.pmc_json:
            key = {"config": line["config"].get("baseline_config", "c2"), "steps": line["steps"], "warmup": line["warmup"]}
            try:
                entries = json.load(open(a.pmc_json))
            except (OSError, ValueError):
                entries = []
            extra = json.loads(sys.stdin.read()) if not sys.stdin.isatty() else {}
            key.update(extra)
            entries = [e for e in entries if e.get("command_key") != key]
            entries.append({"command_key": key, "command": a.command, "launches": nf, "fetch_size_kb_sum": fv, "write_size_kb_sum": wv,
                            "bytes_per_launch": bpl, "source_file": "gemm_pmc_traffic.json",
                            "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads, MI355X_MICROARCH.md HBM section); counters are in KB; "
                                          "Infinity-Cache hits are counted; GEMM launches of the roofline pass only"})
            json.dump(entries, open(a.pmc_json, "w"), indent=1)


if __name__ == "__main__":
    main()
