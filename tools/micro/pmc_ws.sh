#!/bin/bash
# PMC passes over the W-stationary GEMM micro benchmark (M = 15872; N = 256, 768, 2048): where do the waves wait?
# Usage (GPU box, repo root): bash tools/micro/pmc_ws.sh > gpurun_out/pmc_ws.txt   (build tools/micro/gemm_ws.bin first)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM" \
           "TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES" "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc_ws_$i
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_ws_$i -o p --output-format csv -- $ROOT/tools/micro/gemm_ws.bin 15872 > /dev/null 2>/tmp/pmc_ws_$i.err || { echo "pass $i ($set) failed: $(tail -2 /tmp/pmc_ws_$i.err)"; continue; }
  python3 - /tmp/pmc_ws_$i <<'PY'
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(cc)):
    if "k_ws" not in r["Kernel_Name"]:
        continue
    key = (int(r["Grid_Size"]) // 256 if "Grid_Size" in r else int(r.get("Grid_Size_X", 0)) // 256)
    a = acc[key][r["Counter_Name"]]
    a[0] += 1; a[1] += float(r["Counter_Value"])
for key in sorted(acc):
    print(f"  {key} workgroups: " + ", ".join(f"{c} {v[1] / v[0]:.3g}" for c, v in sorted(acc[key].items())) + f"  (avg per launch over {next(iter(acc[key].values()))[0]})")
PY
done
