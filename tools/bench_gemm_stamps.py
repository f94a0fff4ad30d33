"""Phase stamps of k_gemm4's steady-state tile pair (development aid).  Usage: python tools/bench_gemm_stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import translation_transformer_amd as tta
from tests.util_models import tiny_state
st, cfg = tiny_state()
m = tta.NativeTransformer(st, cfg["num_heads"], 0, device=0)
for M, N, K, S in ((2048, 2048, 256, 0), (4096, 2048, 256, 0), (8192, 2048, 256, 0), (8192, 256, 2048, 1), (8192, 256, 2048, 2), (8192, 256, 2048, 4)):
    us, diff = C.c_double(), C.c_double()
    print(f"M={M} N={N} K={K} S={S}", flush=True)
    sys.stderr.flush()
    rc = m._lib.ttx_debug_gemm_bench(m.session, M, N, K, S, 14, 20, C.byref(us), C.byref(diff))
    print(f"   rc={rc} {us.value:.1f} us/launch", flush=True)
