#!/bin/bash
# bench.py --timed-only under environment switches given as arguments ("A=1 B=2" each; "default" = none), for the driver's
# command (c2, 20 steps) and c3.  Usage (GPU box): bash tools/sweep_env.sh "default" "TTX_SMALL_ROWS=0" ...
cd "$(dirname "$0")/.."
for cfg in "$@"; do
  if [ "$cfg" = "default" ]; then e="TTX_NOP=1"; else e="$cfg"; fi
  for what in "--steps 20 --warmup 5" "--config c3"; do
    env $e timeout -k 10 300 python bench.py $what --timed-only --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('[$cfg] bench.py $what ->', round(d['value'],1), 'reactions/s', [round(v) for v in d['repeats']['values']])" || exit 1
  done
done
