# bench.py --timed-only at several list lengths (and optional env switches given as arguments: "A=1 B=2")
for STEPS in 20 128 256; do
  for cfg in "default" "$@"; do
    if [ "$cfg" = "default" ]; then e="TTX_NOP=1"; else e="$cfg"; fi
    env $e timeout -k 10 200 python bench.py --steps $STEPS --warmup 5 --timed-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('steps $STEPS [$cfg] value', round(d['value'],1), 'ms', round(d['ms_per_step']*d['steps'],1))"
  done
done
