# A/B of compile-time GEMM experiments: bash tools/ab_gemm.sh "-DFLAG1" "-DFLAG2" ...   (first run = no flag)
for flags in "" "$@"; do
  echo "== flags: [$flags]"
  TTX_HIPCC_FLAGS="$flags" python -c "import translation_transformer_amd as t; t.build(force=True)" > /dev/null 2>&1 || { echo build failed; continue; }
  timeout -k 10 300 python tools/bench_gemm.py 15872 7936 2>&1 | grep -v "amdgpu.ids" | grep -v "FFN2 S[148]" | cut -c1-250
done
python -c "import translation_transformer_amd as t; t.build(force=True)" > /dev/null 2>&1
