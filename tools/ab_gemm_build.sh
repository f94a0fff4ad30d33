#!/bin/bash
# A/B of compile-time GEMM experiments in isolation: bash tools/ab_gemm_build.sh "" "-DFLAG" ...   ("" = the default build).
# For every flag set: rebuild, then tools/bench_gemm.py on the row counts in $ROWS (difference to the 64x64 result must stay 0.0).
cd "$(dirname "$0")/.."
for flags in "$@"; do
  echo "== flags: [$flags]"
  TTX_HIPCC_FLAGS="$flags" python -c "import translation_transformer_amd as t; t.build(force=True)" > /dev/null 2>&1 || { echo build failed; continue; }
  timeout -k 10 300 python tools/bench_gemm.py ${ROWS:-15872 7936 4960} 2>/dev/null | grep -v amdgpu.ids | sed 's/(diff 0.0e+00)//g' || exit 1
done
python -c "import translation_transformer_amd as t; t.build(force=True)" > /dev/null 2>&1
