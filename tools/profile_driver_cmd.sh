#!/bin/bash
# rocprofv3 evidence for one bench.py configuration: kernel trace of the timed-only run, kernel trace of the run WITH bench.py's
# roofline pass, and the two PMC passes (FETCH_SIZE, WRITE_SIZE) of the latter.  The profiled command is the driver's, minus the
# legs that would follow the roofline pass (tools/roofline_from_trace.py finds the pass as the last GEMM launches of the process):
#   c2:  python3 bench.py --gpus 1 --steps STEPS --warmup WARMUP --no-cpu-baseline --no-sub-records
#   c3:  python3 bench.py --gpus 1 --config c3 --no-cpu-baseline          (c4 likewise)
# Run on the GPU box from the repo root:  bash tools/profile_driver_cmd.sh CONFIG [STEPS] [WARMUP] [TAG]
# Raw output goes to gpurun_out/<TAG>_*; the summaries land in gpurun_out/<TAG>_summary/ (copy them to profiles/).
set -o pipefail
CFG=${1:-c2}; STEPS=${2:-20}; WARM=${3:-5}; TAG=${4:-r03_${CFG}}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$CFG" = "c2" ]; then
  ARGS="--gpus 1 --steps $STEPS --warmup $WARM"
  FULL="$ARGS --no-cpu-baseline --no-sub-records"
  KEY="{\"schedule\": \"rows\", \"inflight\": 8, \"batch_size\": 32, \"n_drafts\": 3, \"draft_len\": 10, \"max_len\": 200}"
elif [ "$CFG" = "c3" ]; then
  ARGS="--gpus 1 --config c3"
  FULL="$ARGS --no-cpu-baseline"
  KEY="{\"schedule\": \"rows\", \"inflight\": 8, \"batch_size\": 4, \"n_drafts\": 7, \"draft_len\": 10, \"max_len\": 200, \"smart\": 0}"
else
  ARGS="--gpus 1 --config c4"
  FULL="$ARGS --no-cpu-baseline"
  KEY="{\"schedule\": \"rows\", \"inflight\": 8, \"batch_size\": 8, \"n_drafts\": 2, \"draft_len\": 10, \"max_len\": 200, \"smart\": 0}"
fi
echo "[1/4] kernel trace, timed-only"
timeout -k 10 ${TRACE_TIMEOUT:-400} rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_timed -o t --output-format csv -- python3 $ROOT/bench.py $ARGS --timed-only --repeats 2 > $OUT/${TAG}_timed.jsonl 2> $OUT/${TAG}_timed.err || exit 1
echo "[2/4] kernel trace, with the roofline pass"
timeout -k 10 ${TRACE_TIMEOUT:-400} rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_full -o f --output-format csv -- python3 $ROOT/bench.py $FULL > $OUT/${TAG}_full.jsonl 2> $OUT/${TAG}_full.err || exit 1
echo "[3/4] PMC FETCH_SIZE"
timeout -k 10 ${PMC_TIMEOUT:-700} rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_fetch -o p --output-format csv -- python3 $ROOT/bench.py $FULL > $OUT/${TAG}_fetch.jsonl 2> $OUT/${TAG}_fetch.err || exit 1
echo "[4/4] PMC WRITE_SIZE"
timeout -k 10 ${PMC_TIMEOUT:-700} rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_write -o p --output-format csv -- python3 $ROOT/bench.py $FULL > $OUT/${TAG}_write.jsonl 2> $OUT/${TAG}_write.err || exit 1
# summaries (small) next to the raw output; the raw CSVs are too large to travel back (64 MiB cap) and are removed
S=$OUT/${TAG}_summary
mkdir -p $S
cd $ROOT
python3 tools/prof_summary.py $OUT/${TAG}_timed/t_kernel_trace.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py $ARGS --timed-only --repeats 2  (whole process: set-up, warm-up and two repeats of the timed region)" > $S/timed_only_kernel_trace.txt
python3 tools/trace_timeline.py $OUT/${TAG}_timed/t_kernel_trace.csv 1 12 > $S/timed_only_timeline.txt
cp $OUT/${TAG}_timed/t_kernel_stats.csv $S/timed_only_kernel_stats.csv
cp $OUT/${TAG}_full/f_kernel_stats.csv $S/full_kernel_stats.csv
cp $OUT/${TAG}_timed.jsonl $OUT/${TAG}_full.jsonl $OUT/${TAG}_fetch.jsonl $OUT/${TAG}_write.jsonl $S/
cp $ROOT/profiles/gemm_pmc_traffic.json $S/gemm_pmc_traffic.json 2>/dev/null
echo "$KEY" | python3 tools/roofline_from_trace.py $OUT/${TAG}_full/f_kernel_trace.csv $OUT/${TAG}_full.jsonl \
  --fetch $OUT/${TAG}_fetch/p_counter_collection.csv $OUT/${TAG}_fetch/p_kernel_trace.csv \
  --write $OUT/${TAG}_write/p_counter_collection.csv $OUT/${TAG}_write/p_kernel_trace.csv \
  --pmc-json $S/gemm_pmc_traffic.json \
  --command "rocprofv3 --kernel-trace [--stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE] -- python3 bench.py $FULL" \
  > $S/roofline_pass_from_trace.txt || exit 1
rm -rf $OUT/${TAG}_timed $OUT/${TAG}_full $OUT/${TAG}_fetch $OUT/${TAG}_write
ls -la $S
