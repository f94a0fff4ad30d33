"""tools/roofline_from_trace.py on a small synthetic rocprofv3 trace: the roofline pass is the LAST roofline.launches GEMM
launches of the process (it runs on the timed region's own pools, i.e. on several streams; the same number of launches before it
is its warm run), its average launch duration and achieved TFLOP/s are recomputed from the trace, and the PMC passes are reduced
to bytes per launch for exactly those launches."""
import csv
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
TRACE_COLS = ["Kind", "Agent_Id", "Queue_Id", "Stream_Id", "Thread_Id", "Dispatch_Id", "Kernel_Id", "Kernel_Name", "Correlation_Id",
              "Start_Timestamp", "End_Timestamp", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
              "Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"]


def _trace(path, rows):
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=TRACE_COLS)
        w.writeheader()
        for i, (stream, name, start, dur) in enumerate(rows):
            w.writerow({c: 0 for c in TRACE_COLS} | {"Kind": "KERNEL_DISPATCH", "Stream_Id": stream, "Dispatch_Id": i + 1, "Kernel_Name": name,
                                                      "Start_Timestamp": start, "End_Timestamp": start + dur, "Workgroup_Size_X": 256,
                                                      "Workgroup_Size_Y": 1, "Workgroup_Size_Z": 1, "Grid_Size_X": 1024, "Grid_Size_Y": 4,
                                                      "Grid_Size_Z": 1})


def _counters(path, rows, counter, value):
    cols = ["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Grid_Size", "Kernel_Id", "Kernel_Name",
            "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name",
            "Counter_Value", "Start_Timestamp", "End_Timestamp"]
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=cols)
        w.writeheader()
        for i, (stream, name, start, dur) in enumerate(rows):
            w.writerow({c: 0 for c in cols} | {"Dispatch_Id": i + 1, "Kernel_Name": name, "Counter_Name": counter, "Counter_Value": value})


def test_roofline_pass_is_the_last_launches_and_reduced(tmp_path):
    gemm = "void ttx::k_gemm24<4>(ttx::GemmArgs)"
    rows = []
    for i in range(10):                                   # timed region: 10 GEMM launches of 50 us on two streams
        rows.append((2 + i % 2, gemm, 1000 * i, 50_000))
    for i in range(6):                                    # warm run of the pass: same shapes, 30 us
        rows.append((2 + i % 2, gemm, 100_000 + 1000 * i, 30_000))
    for i in range(6):                                    # roofline pass: 6 GEMM launches of 20 us on two streams + other kernels
        rows.append((2 + i % 2, gemm, 200_000 + 1000 * i, 20_000))
        rows.append((2 + i % 2, "ttx::k_finish_ln<4>(ttx::FinishArgs)", 200_500 + 1000 * i, 5_000))
    _trace(tmp_path / "t.csv", rows)
    _counters(tmp_path / "f_cc.csv", rows, "FETCH_SIZE", 1000.0)      # KB per launch
    _counters(tmp_path / "w_cc.csv", rows, "WRITE_SIZE", 500.0)
    line = {"value": 1.0, "unit": "reactions/s", "steps": 20, "warmup": 5, "config": {"workload": "x"},
            "roofline": {"launches": 6, "avg_launch_us": 21.0, "achieved": 95.0, "unit": "TFLOP/s", "frac": 0.6, "peak": 157.3,
                         "flops_per_launch": 2.0e9, "algorithmic_bytes_per_launch": 1.0e6, "event_pair_overhead_us": 4.7}}
    (tmp_path / "b.jsonl").write_text("noise\n" + json.dumps(line) + "\n")
    out_json = tmp_path / "pmc.json"
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "roofline_from_trace.py"), str(tmp_path / "t.csv"), str(tmp_path / "b.jsonl"),
                        "--fetch", str(tmp_path / "f_cc.csv"), str(tmp_path / "t.csv"), "--write", str(tmp_path / "w_cc.csv"),
                        str(tmp_path / "t.csv"), "--pmc-json", str(out_json), "--command", "test"],
                       input='{"schedule": "rows", "inflight": 8}', capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "roofline pass = the last 6 GEMM launches of the process on streams ['2', '3']" in r.stdout
    assert "have the same (kernel, grid) counts" in r.stdout
    assert "average 20.00 us per launch" in r.stdout
    assert "100.0 TFLOP/s" in r.stdout                    # 6 x 2 GFLOP / 120 us
    entry = json.loads(out_json.read_text())[0]
    assert entry["launches"] == 6 and entry["command_key"] == {"config": "c2", "steps": 20, "warmup": 5, "schedule": "rows", "inflight": 8}
    assert abs(entry["bytes_per_launch"] - (2 * 1000 + 500) * 1024) < 1e-6


def test_timeline_and_pmc_summaries_on_a_synthetic_trace(tmp_path):
    """tools/trace_timeline.py (last busy window with >= 1000 launches, busy share, kernels running at once),
    tools/pmc_by_kernel.py and tools/pmc_top_dispatches.py on small synthetic rocprofv3 CSVs."""
    gemm, ln = "void ttx::k_gemm24<4>(ttx::GemmArgs)", "ttx::k_finish_ln<4>(ttx::FinishArgs)"
    rows = [(1, gemm, 0, 1000)]                                       # an early, small window
    t = 50_000_000
    for i in range(600):                                              # the window of interest: two streams, overlapping halves
        rows.append((2, gemm, t + 2000 * i, 1500))
        rows.append((3, ln, t + 2000 * i + 1000, 1500))
    _trace(tmp_path / "t.csv", rows)
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "trace_timeline.py"), str(tmp_path / "t.csv"), "1", "4"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "1200 launches on 2 streams" in r.stdout
    assert "k_gemm24<4>" in r.stdout and "600 launches" in r.stdout
    body = [l for l in r.stdout.split("\n") if l.strip().startswith(("0 |", "1 |", "2 |", "3 |"))]
    assert len(body) == 4
    for line in body:                                                 # every slice: always busy, 1.5 kernels at once
        cols = [c.strip() for c in line.split("|")]
        assert abs(float(cols[1]) - 1.0) < 0.02 and abs(float(cols[2]) - 1.5) < 0.05, line
    _counters(tmp_path / "cc.csv", rows[:5], "SQ_WAVE_CYCLES", 10.0)
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "pmc_by_kernel.py"), str(tmp_path / "cc.csv"), "ttx"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "ttx::k_gemm24<4> | 3 | 30" in r.stdout and "ttx::k_finish_ln<4> | 2 | 20" in r.stdout
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "pmc_top_dispatches.py"), str(tmp_path / "cc.csv"), "k_gemm24", "2",
                        str(tmp_path / "t.csv")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("ttx::k_gemm24<4>") == 2 and "SQ_WAVE_CYCLES" in r.stdout
