"""GEMM shapes of the verify step in isolation (ttx_debug_gemm_bench): microseconds per launch, TFLOP/s and the largest
absolute difference to the 64x64 tiling's result — which must be 0.0 for every variant: all of them evaluate the canonical
slice sum of translation-transformer_amd/csrc/ttx_gemm.hip.
Variants: 2 = 64x64 tiles, 46 = 128x64, 24 = k_gemm24's own choice from the row count,
3 = one wave per slice (32x32 tiles, K = 256), 8 = one workgroup per slice + slabs (FFN2).
Usage: python tools/bench_gemm.py [M ...]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import translation_transformer_amd as tta
from tests.util_models import tiny_state  # any model gives a session

st, cfg = tiny_state()
m = tta.NativeTransformer(st, cfg["num_heads"], 0, device=0)
lib = m._lib
Ms = [int(x) for x in sys.argv[1:]] or [15872, 7936, 4960, 2480, 992, 310]
shapes = [("FFN1", 2048, 256, (2, 46, 24)), ("QKV", 768, 256, (2, 46, 24, 3)), ("dxd", 256, 256, (2, 46, 24, 3)),
          ("FFN2", 256, 2048, (2, 46, 24, 8))]
worst = 0.0
for M in Ms:
    for name, N, K, variants in shapes:
        row = f"M={M:5d} {name:5s} N={N:4d} K={K:4d}:"
        for variant in variants:
            us, diff = C.c_double(), C.c_double()
            S = 8 if variant == 8 else 0
            rc = lib.ttx_debug_gemm_bench(m.session, M, N, K, S, variant, 50, C.byref(us), C.byref(diff))
            if rc:
                row += f"  v{variant}: rc={rc} {lib.ttx_last_error().decode()[:60]}"
                continue
            tf = 2.0 * M * N * K / (us.value * 1e-6) / 1e12
            worst = max(worst, diff.value)
            row += f"  v{variant}: {us.value:7.1f} us {tf:6.1f} TF/s (diff {diff.value:.1e})"
        print(row, flush=True)
print("largest difference between any two variants:", worst)
sys.exit(0 if worst == 0.0 else 1)
