#!/bin/bash
# rocprofv3 evidence for the command the driver runs (python3 bench.py --gpus 1 --steps 20 --warmup 5): kernel trace of the
# timed-only run, kernel trace of the run WITH bench.py's roofline pass, and the two PMC passes (FETCH_SIZE, WRITE_SIZE) of the
# latter.  Run on the GPU box from the repo root:  bash tools/profile_driver_cmd.sh [STEPS] [WARMUP] [TAG]
# Raw output goes to gpurun_out/<TAG>_*; summaries are written under profiles/ by tools/roofline_from_trace.py afterwards.
set -o pipefail
STEPS=${1:-20}; WARM=${2:-5}; TAG=${3:-r02_s${STEPS}}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="$ROOT/bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-cpu-baseline"
echo "[1/4] kernel trace, timed-only"
timeout -k 10 ${TRACE_TIMEOUT:-300} rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_timed -o t --output-format csv -- python3 $ROOT/bench.py --gpus 1 --steps $STEPS --warmup $WARM --timed-only > $OUT/${TAG}_timed.jsonl 2> $OUT/${TAG}_timed.err || exit 1
echo "[2/4] kernel trace, with the roofline pass"
timeout -k 10 ${TRACE_TIMEOUT:-300} rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_full -o f --output-format csv -- python3 $CMD > $OUT/${TAG}_full.jsonl 2> $OUT/${TAG}_full.err || exit 1
echo "[3/4] PMC FETCH_SIZE"
timeout -k 10 ${PMC_TIMEOUT:-600} rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_fetch -o p --output-format csv -- python3 $CMD > $OUT/${TAG}_fetch.jsonl 2> $OUT/${TAG}_fetch.err || exit 1
echo "[4/4] PMC WRITE_SIZE"
timeout -k 10 ${PMC_TIMEOUT:-600} rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_write -o p --output-format csv -- python3 $CMD > $OUT/${TAG}_write.jsonl 2> $OUT/${TAG}_write.err || exit 1
ls -la $OUT/${TAG}_*/ | head -40
# summaries (small) next to the raw output; the raw CSVs are too large to travel back (64 MiB cap) and are removed
S=$OUT/${TAG}_summary
mkdir -p $S
cd $ROOT
python3 tools/prof_summary.py $OUT/${TAG}_timed/t_kernel_trace.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps $STEPS --warmup $WARM --timed-only  (whole process: warm-up + timed region)" > $S/timed_only_kernel_trace.txt
cp $OUT/${TAG}_timed/t_kernel_stats.csv $S/timed_only_kernel_stats.csv
cp $OUT/${TAG}_full/f_kernel_stats.csv $S/full_kernel_stats.csv
cp $OUT/${TAG}_timed.jsonl $OUT/${TAG}_full.jsonl $OUT/${TAG}_fetch.jsonl $OUT/${TAG}_write.jsonl $S/
echo "{\"schedule\": \"rows\", \"inflight\": 8, \"batch_size\": 32, \"n_drafts\": 3, \"draft_len\": 10, \"max_len\": 200}" | \
python3 tools/roofline_from_trace.py $OUT/${TAG}_full/f_kernel_trace.csv $OUT/${TAG}_full.jsonl \
  --fetch $OUT/${TAG}_fetch/p_counter_collection.csv $OUT/${TAG}_fetch/p_kernel_trace.csv \
  --write $OUT/${TAG}_write/p_counter_collection.csv $OUT/${TAG}_write/p_kernel_trace.csv \
  --pmc-json $S/gemm_pmc_traffic_entry.json \
  --command "rocprofv3 --kernel-trace [--stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE] -- python3 bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-cpu-baseline" \
  > $S/roofline_pass_from_trace.txt || exit 1
rm -rf $OUT/${TAG}_timed $OUT/${TAG}_full $OUT/${TAG}_fetch $OUT/${TAG}_write
ls -la $S
