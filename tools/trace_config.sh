#!/bin/bash
# Kernel trace of one bench.py configuration's timed-only run, reduced on the box to the timeline / per-kernel summary of the
# last repeat of the timed region (tools/trace_timeline.py).  Usage (GPU box, repo root): bash tools/trace_config.sh c3|c4|c2 TAG [bench args]
CFG=$1; TAG=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 ${TRACE_TIMEOUT:-500} rocprofv3 --kernel-trace -d $OUT/${TAG}_trace -o t --output-format csv -- python3 $ROOT/bench.py --config $CFG --timed-only --repeats 2 "$@" > $OUT/${TAG}.jsonl 2> $OUT/${TAG}.err || exit 1
python3 $ROOT/tools/trace_timeline.py $OUT/${TAG}_trace/t_kernel_trace.csv 1 12 > $OUT/${TAG}_timeline.txt || exit 1
rm -rf $OUT/${TAG}_trace
cat $OUT/${TAG}_timeline.txt
