#!/bin/bash
# bench.py --config $CFG --timed-only under environment switches given as arguments ("A=1 B=2" each; "default" = none).
# Usage (GPU box): CFG=c3 bash tools/sweep_beam.sh "default" "TTX_POOL_SESSIONS=2 TTX_BEAM_POOL_CAPACITY=92" ...
cd "$(dirname "$0")/.."
for cfg in "$@"; do
  if [ "$cfg" = "default" ]; then e="TTX_NOP=1"; else e="$cfg"; fi
  env $e timeout -k 10 400 python bench.py --config ${CFG:-c3} --timed-only --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('[${CFG:-c3}: $cfg] ->', round(d['value'],1), 'reactions/s', [round(v) for v in d['repeats']['values']], 'device iterations', d.get('device_iterations_rank0'))" || exit 1
done
