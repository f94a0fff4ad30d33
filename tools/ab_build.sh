#!/bin/bash
# A/B of compile-time experiments: bash tools/ab_build.sh "" "-DFLAG1" "-DFLAG2 -DX=3" ...   ("" = the default build).
# For every flag set: rebuild, then bench.py --timed-only on the driver's command, the 256-batch list, c3 and the one-batch-at-a-time loop (3 repeats each).
cd "$(dirname "$0")/.."
for flags in "$@"; do
  echo "== flags: [$flags]"
  TTX_HIPCC_FLAGS="$flags" python -c "import translation_transformer_amd as t; t.build(force=True)" > /dev/null 2>&1 || { echo build failed; continue; }
  for what in "--steps 20 --warmup 5" "--steps 256" "--config c3" "--steps 20 --warmup 5 --schedule batches --inflight 1" ${AB_EXTRA:+"$AB_EXTRA"}; do
    timeout -k 10 300 python bench.py $what --timed-only --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('   bench.py $what ->', round(d['value'],1), 'reactions/s', [round(v) for v in d['repeats']['values']])" || exit 1
  done
done
python -c "import translation_transformer_amd as t; t.build(force=True)" > /dev/null 2>&1
