#!/bin/bash
# A/B of the beam paths' small-row GEMM policy (which kernel takes the narrow step GEMMs, split-K factors) on C3, 8 batches in
# flight and one at a time.  Usage (GPU box): bash tools/ab_beam_policy.sh > gpurun_out/ab_beam_policy.txt
cd "$(dirname "$0")/.."
run() { env "$@" python bench.py --config c3 --timed-only $EXTRA 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.readlines()[-1])['value'],1))"; }
for cfg in "TTX_FFN2_G2=0" "TTX_FFN2_G2=1" "TTX_FFN2_G2=1 TTX_TREE_GEMM3_MAX_N=256" "TTX_FFN2_G2=1 TTX_TREE_GEMM3_MAX_N=0" \
           "TTX_FFN2_G2=1 TTX_TREE_GEMM3_MAX_N=0 TTX_TREE_PROJ_SPLIT=2" "TTX_FFN2_G2=1 TTX_TREE_GEMM3_MAX_N=0 TTX_TREE_PROJ_SPLIT=1" \
           "TTX_FFN2_G2=1 TTX_TREE_FFN2_SPLIT=4" "TTX_FFN2_G2=1 TTX_TREE_FFN2_SPLIT=2" "TTX_FFN2_G2=1 TTX_TREE_GEMM3_MAX_N=0 TTX_TREE_PROJ_SPLIT=1 TTX_TREE_FFN2_SPLIT=4"; do
  EXTRA="" ; a=$(run $cfg); EXTRA="--inflight 1"; b=$(run $cfg)
  echo "$cfg: 8 in flight $a, one at a time $b reactions/s"
done
