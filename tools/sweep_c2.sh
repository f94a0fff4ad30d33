#!/bin/bash
# bench.py --steps 20 --warmup 5 --timed-only (the driver's c2 command) under environment switches given as arguments
# ("A=1 B=2" each; "default" = none).  Usage (GPU box): bash tools/sweep_c2.sh "default" "TTX_POOL_SESSIONS=2" ...
cd "$(dirname "$0")/.."
for cfg in "$@"; do
  if [ "$cfg" = "default" ]; then e="TTX_NOP=1"; else e="$cfg"; fi
  env $e timeout -k 10 300 python bench.py --steps ${STEPS:-20} --warmup 5 --timed-only --repeats 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('[$cfg] ->', round(d['value'],1), 'reactions/s', [round(v) for v in d['repeats']['values']], 'device steps', d.get('device_model_calls'))" || exit 1
done
