# experiments on short work lists: bench.py --steps $1 --warmup 5 under kernel-policy switches
STEPS=${1:-20}
for cfg in "default" "TTX_FUSE_LN_MIN_ROWS=1" "TTX_FUSE_LN_MIN_ROWS=1 TTX_FFN2_SPLIT=1" "TTX_FFN2_SPLIT=1" "TTX_BIG_MIN_TILES=96" "TTX_BIG_MIN_TILES=400" "TTX_ATTN_SPLIT=0"; do
  echo "$cfg"
  if [ "$cfg" = "default" ]; then cfg="TTX_NOP=1"; fi
  env $cfg timeout -k 10 120 python bench.py --steps $STEPS --warmup 5 --timed-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('  value', round(d['value'],1), 'ms', round(d['ms_per_step']*d['steps'],1))"
done
