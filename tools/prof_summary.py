"""Per-kernel and per-launch-shape summary of a rocprofv3 --kernel-trace run (rocpd .db or *_kernel_trace.csv).
Usage: python tools/prof_summary.py <results.db | kernel_trace.csv> ["title line"]"""
import csv
import sqlite3
import statistics
import sys
from collections import defaultdict


def rows_from_db(path):
    c = sqlite3.connect(path)
    q = "select name, start, end, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z, lds_size, vgpr_count from kernels"
    for name, st, en, gx, gy, gz, wx, wy, wz, lds, vg in c.execute(q):
        yield name, en - st, (gx // max(wx, 1), gy // max(wy, 1), gz // max(wz, 1)), lds, vg


def rows_from_csv(path):
    with open(path) as f:
        for r in csv.DictReader(f):
            g = tuple(int(r[f"Grid_Size_{a}"]) // max(1, int(r[f"Workgroup_Size_{a}"])) for a in "XYZ")
            yield r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), g, int(r.get("LDS_Block_Size", 0)), int(r.get("VGPR_Count", 0))


def main():
    path = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else path
    rows = list(rows_from_db(path) if path.endswith(".db") else rows_from_csv(path))
    per_kernel, per_shape = defaultdict(list), defaultdict(list)
    for name, dur, grid, lds, vg in rows:
        short = name.split("(")[0].replace("void ", "")
        per_kernel[short].append(dur)
        per_shape[(short, grid, lds, vg)].append(dur)
    total = sum(sum(v) for v in per_kernel.values())
    print(title)
    print(f"total kernel time {total / 1e6:.1f} ms over {len(rows)} launches")
    print("\nkernel | launches | total ms | share | avg us | median us | min us | max us")
    for k, v in sorted(per_kernel.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k} | {len(v)} | {sum(v) / 1e6:.1f} | {100 * sum(v) / total:.1f}% | {sum(v) / len(v) / 1e3:.2f} | "
              f"{statistics.median(v) / 1e3:.2f} | {min(v) / 1e3:.2f} | {max(v) / 1e3:.2f}")
    print("\nkernel | grid (wg x,y,z) | LDS B | VGPR | launches | total ms | avg us | median us | min us | max us")
    shown = 0
    for (k, grid, lds, vg), v in sorted(per_shape.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k} | {grid[0]},{grid[1]},{grid[2]} | {lds} | {vg} | {len(v)} | {sum(v) / 1e6:.1f} | {sum(v) / len(v) / 1e3:.2f} | "
              f"{statistics.median(v) / 1e3:.2f} | {min(v) / 1e3:.2f} | {max(v) / 1e3:.2f}")
        shown += 1
        if shown >= 40:
            break


if __name__ == "__main__":
    main()
