"""CPU-side checks of the drop-in boundary: the library builds for gfx950, loads, exports exactly the
entry points include/ttx.h declares, and refuses to run without an MI355X (no CPU fallback)."""
import ctypes as C
import re
from pathlib import Path

import pytest
import torch

import translation_transformer_amd as tta
from translation_transformer_amd import _native as N

ROOT = Path(__file__).resolve().parent.parent


def declared_functions():
    text = (ROOT / "include" / "ttx.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ttx_[a-z_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_functions() == sorted(N.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = tta.lib()
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert lib.ttx_abi_version() == 4


def test_struct_sizes_match_header_layout():
    assert C.sizeof(N.Config) == 10 * 4
    assert C.sizeof(N.GenParams) == 8 * 4
    assert C.sizeof(N.GenStats) == 6 * 8 + 2 * 8 + 2 * 8
    assert C.sizeof(N.Tensor) == 24


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a machine without a GPU")
def test_no_cpu_fallback():
    lib = tta.lib()
    assert lib.ttx_device_count() == 0
    cfg = N.Config(30, 30, 64, 2, 128, 2, 2, 0, 5000, 1e-5)
    model = C.c_void_p()
    rc = lib.ttx_model_create_empty(C.byref(cfg), 0, C.byref(model))
    assert rc == N.TTX_ERR_NO_DEVICE and b"no CPU fallback" in lib.ttx_last_error()
    with pytest.raises(RuntimeError):
        tta.NativeTransformer({}, 2)
    with pytest.raises(TypeError):
        tta.TranslationInferenceGreedySpeculative(object(), 10, 3, 1, 0, 1, 2, 4)
