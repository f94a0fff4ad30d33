#!/bin/bash
# Kernel trace of the driver's c2 command with ONE slot pool of 640 slots (no other stream running beside it: durations
# are the kernels' own), reduced on the box to tools/prof_summary.py's per-kernel / per-shape table.
# Usage (GPU box, repo root): bash tools/trace_single_pool.sh TAG
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export TTX_POOL_SESSIONS=1 TTX_POOL_CAPACITY=640
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/${TAG}_trace -o t --output-format csv -- python3 $ROOT/bench.py --steps 20 --warmup 5 --timed-only --repeats 2 > $OUT/${TAG}.jsonl 2> $OUT/${TAG}.err || exit 1
python3 $ROOT/tools/prof_summary.py $OUT/${TAG}_trace/t_kernel_trace.csv "one pool of 640 slots, driver command, whole process" > $OUT/${TAG}_summary.txt || exit 1
python3 $ROOT/tools/trace_timeline.py $OUT/${TAG}_trace/t_kernel_trace.csv 1 12 >> $OUT/${TAG}_summary.txt
rm -rf $OUT/${TAG}_trace
