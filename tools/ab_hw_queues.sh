#!/bin/bash
# A/B of the HIP runtime's hardware-queue pool (GPU_MAX_HW_QUEUES, default 4: streams beyond that share queues) and of the
# number of batches in flight on the beam-speculative path.  Usage (on the GPU box): bash tools/ab_hw_queues.sh [c3|c4]
set -o pipefail
cd "$(dirname "$0")/.."
cfg=${1:-c3}
for q in 1 2 3 4 6; do
  for f in 4 8 16; do
    v=$(GPU_MAX_HW_QUEUES=$q python bench.py --config $cfg --steps 32 --inflight $f --timed-only 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readlines()[-1])['value'])") || exit 1
    echo "GPU_MAX_HW_QUEUES=$q bench.py --config $cfg --steps 32 --inflight $f --timed-only -> $v reactions/s"
  done
done
