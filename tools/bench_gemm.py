"""GEMM shapes of the verify step in isolation (ttx_debug_gemm_bench): microseconds per launch and TFLOP/s of the
64x64 (2), 32x32 (3) and 128x128 (4) kernels.  Usage: python tools/bench_gemm.py [M ...]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import translation_transformer_amd as tta
from tests.util_models import tiny_state  # any model gives a session

st, cfg = tiny_state()
m = tta.NativeTransformer(st, cfg["num_heads"], 0, device=0)
lib = m._lib
Ms = [int(x) for x in sys.argv[1:]] or [7936, 4800, 992]
shapes = [("FFN1", 2048, 256, 0), ("QKV", 768, 256, 0), ("dxd", 256, 256, 1), ("FFN2 S1", 256, 2048, 1), ("FFN2 S2", 256, 2048, 2),
          ("FFN2 S4", 256, 2048, 4), ("FFN2 S8", 256, 2048, 8)]
for M in Ms:
    for name, N, K, S in shapes:
        row = f"M={M:5d} {name:8s} N={N:4d} K={K:4d} S={S}:"
        for variant in (2, 4, 46, 24, 66):
            if variant in (24, 66) and (K // max(S, 1)) % 256:
                continue
            us, diff = C.c_double(), C.c_double()
            rc = lib.ttx_debug_gemm_bench(m.session, M, N, K, S, variant, 50, C.byref(us), C.byref(diff))
            if rc:
                row += f"  v{variant}: rc={rc}"
                continue
            tf = 2.0 * M * N * K / (us.value * 1e-6) / 1e12
            row += f"  v{variant}: {us.value:7.1f} us {tf:6.1f} TF/s (diff {diff.value:.1e})"
        print(row, flush=True)
