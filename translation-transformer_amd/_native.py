"""ctypes binding of libttx_hip.so (include/ttx.h) + the in-tree build recipe.

The shared library is the product; this module only loads it and marshals pointers.  There is no
fallback: if the library is missing, cannot be loaded, or finds no gfx950 device, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
INCLUDE = PKG_DIR.parent / "include"
LIB_PATH = PKG_DIR / "libttx_hip.so"
# translation units (compiled in parallel, linked into one shared library) and the headers every one of them depends on
UNITS = [CSRC / "ttx_api.hip", CSRC / "ttx_gemm.hip", CSRC / "ttx_attn.hip"]
HEADERS = [CSRC / "ttx_internal.h", CSRC / "ttx_common.hip.h", CSRC / "ttx_loop_kernels.hip.h", CSRC / "ttx_select.h",
           CSRC / "ttx_tokenizer.h", INCLUDE / "ttx.h"]
SOURCES = UNITS + HEADERS
OBJ_DIR = CSRC / "build"

TTX_OK, TTX_ERR_INVALID, TTX_ERR_HIP, TTX_ERR_NO_DEVICE, TTX_ERR_REFERENCE, TTX_ERR_NOMEM = 0, -1, -2, -3, -4, -5
TTX_ERR_ROW_REPLAY = -6
TTX_ERR_MAX_STEPS = -7


class TtxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libttx_hip error {code}: {msg}")
        self.code = code


class ReferenceError_(RuntimeError):
    """Raised where the reference implementation itself raises on the same input."""


class Config(C.Structure):
    _fields_ = [("vocab_size", C.c_int32), ("src_vocab_size", C.c_int32), ("embedding_dim", C.c_int32),
                ("num_heads", C.c_int32), ("feedforward_dim", C.c_int32), ("num_encoder_layers", C.c_int32),
                ("num_decoder_layers", C.c_int32), ("pad_token", C.c_int32), ("max_positions", C.c_int32),
                ("layer_norm_eps", C.c_float)]


class Tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("numel", C.c_int64)]


class GenParams(C.Structure):
    _fields_ = [("max_len", C.c_int32), ("draft_len", C.c_int32), ("n_drafts", C.c_int32), ("pad_token", C.c_int32),
                ("bos_token", C.c_int32), ("eos_token", C.c_int32), ("replace_token", C.c_int32),
                ("want_logits", C.c_int32)]


class BeamParams(C.Structure):
    _fields_ = [("max_len", C.c_int32), ("n_best", C.c_int32), ("draft_len", C.c_int32), ("n_drafts", C.c_int32),
                ("smart_drafts_mode", C.c_int32), ("pad_token", C.c_int32), ("bos_token", C.c_int32),
                ("eos_token", C.c_int32), ("replace_token", C.c_int32), ("max_steps", C.c_int32)]


class BeamStats(C.Structure):
    _fields_ = [("model_calls", C.c_int64), ("input_lines", C.c_int64), ("running_rows", C.c_int64),
                ("accepted_tokens", C.c_int64), ("produced_non_pad_tokens", C.c_int64), ("verified_positions", C.c_int64),
                ("executed_positions", C.c_int64), ("kv_prefix_positions", C.c_int64), ("running_candidates", C.c_int64),
                ("src_tokens_padded", C.c_int64), ("encode_ms", C.c_double), ("decode_ms", C.c_double),
                ("out_width", C.c_int32), ("status", C.c_int32)]


class BeamSearchParams(C.Structure):
    _fields_ = [("max_len", C.c_int32), ("beam_size", C.c_int32), ("pad_token", C.c_int32), ("bos_token", C.c_int32),
                ("eos_token", C.c_int32)]


class BeamSearchStats(C.Structure):
    _fields_ = [("model_calls", C.c_int64), ("running_rows", C.c_int64), ("out_width", C.c_int32), ("pad_", C.c_int32)]


class GenStats(C.Structure):
    _fields_ = [("model_calls", C.c_int64), ("accepted_tokens", C.c_int64), ("produced_tokens", C.c_int64),
                ("verified_positions", C.c_int64), ("kv_prefix_positions", C.c_int64), ("src_positions", C.c_int64),
                ("encode_ms", C.c_double), ("decode_ms", C.c_double), ("src_tokens_padded", C.c_int64),
                ("status", C.c_int64)]


# every symbol include/ttx.h declares: (name, restype, argtypes)
_VP, _I, _I64P = C.c_void_p, C.c_int, C.c_void_p
SYMBOLS = {
    "ttx_abi_version": (C.c_int, []),
    "ttx_last_error": (C.c_char_p, []),
    "ttx_device_count": (C.c_int, []),
    "ttx_model_create": (C.c_int, [C.POINTER(Config), C.POINTER(Tensor), _I, _I, C.POINTER(_VP)]),
    "ttx_model_destroy": (None, [_VP]),
    "ttx_model_create_empty": (C.c_int, [C.POINTER(Config), _I, C.POINTER(_VP)]),
    "ttx_model_blob": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(C.c_int64)]),
    "ttx_session_create": (C.c_int, [_VP, C.POINTER(_VP)]),
    "ttx_session_destroy": (None, [_VP]),
    "ttx_encode_src": (C.c_int, [_VP, _VP, _I, _I, _VP, _VP]),
    "ttx_decode_tgt": (C.c_int, [_VP, _VP, _I, _I, _VP, _VP, _VP, _I, _I, _VP, _VP]),
    "ttx_forward": (C.c_int, [_VP, _VP, _I, _I, _VP, _I, _VP, _VP]),
    "ttx_make_drafts": (C.c_int, [_VP, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _I, _VP, _VP]),
    "ttx_greedy_speculative_generate": (C.c_int, [_VP, _VP, _I, _I, C.POINTER(GenParams), _VP, C.POINTER(GenStats), _VP]),
    "ttx_greedy_generate": (C.c_int, [_VP, _VP, _I, _I, C.POINTER(GenParams), _VP, C.POINTER(GenStats), _VP]),
    "ttx_greedy_speculative_generate_many": (C.c_int, [C.POINTER(_VP), _I, _I, C.POINTER(_VP), C.POINTER(C.c_int),
                                                      C.POINTER(C.c_int), C.POINTER(GenParams), C.POINTER(_VP),
                                                      C.POINTER(GenStats), _VP]),
    "ttx_greedy_speculative_generate_rows": (C.c_int, [C.POINTER(_VP), _I, _I, C.POINTER(_VP), C.POINTER(C.c_int),
                                                      C.POINTER(C.c_int), C.POINTER(GenParams), C.POINTER(_VP),
                                                      C.POINTER(_VP), C.POINTER(_VP), C.POINTER(GenStats), _VP]),
    "ttx_greedy_speculative_generate_pool": (C.c_int, [C.POINTER(_VP), _I, _VP, _I, _I, C.POINTER(C.c_int32), _I,
                                                      C.POINTER(GenParams), _VP, _VP, _VP, C.POINTER(GenStats), _VP]),
    "ttx_beam_speculative_generate": (C.c_int, [_VP, _VP, _I, _I, C.POINTER(BeamParams), _VP, C.POINTER(BeamStats), _VP]),
    "ttx_beam_speculative_generate_many": (C.c_int, [C.POINTER(_VP), _I, _I, C.POINTER(_VP), C.POINTER(C.c_int),
                                                    C.POINTER(C.c_int), C.POINTER(BeamParams), C.POINTER(_VP),
                                                    C.POINTER(BeamStats), _VP]),
    "ttx_beam_speculative_generate_pool": (C.c_int, [C.POINTER(_VP), _I, _VP, _I, _I, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I,
                                                    C.POINTER(C.c_int32), _I, C.POINTER(BeamParams), _VP, _VP, _VP, _I,
                                                    C.POINTER(BeamStats), _VP]),
    "ttx_beam_generate": (C.c_int, [_VP, _VP, _I, _I, C.POINTER(BeamSearchParams), _VP, C.POINTER(BeamSearchStats), _VP]),
    "ttx_nucleus_mask": (C.c_int, [_VP, _VP, _I, _I, C.c_float, _I, C.c_float, _VP, _VP]),
    "ttx_accepted_lengths": (C.c_int, [_VP, _VP, _VP, _I, _I, _I, C.c_float, _I, _VP, _VP]),
    "ttx_ragged_topk": (C.c_int, [_VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP]),
    "ttx_tokenizer_create": (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(C.c_int32), _I, C.POINTER(_VP)]),
    "ttx_tokenizer_destroy": (None, [_VP]),
    "ttx_tokenizer_encode": (C.c_int, [_VP, C.c_char_p, _VP, _I]),
    "ttx_tokenizer_encode_batch": (C.c_int, [_VP, C.POINTER(C.c_char_p), _I, _VP, _I]),
    "ttx_tokenizer_decode": (C.c_int, [_VP, _VP, _I, _VP, _I]),
    "ttx_debug_step_snapshot": (C.c_int, [_VP, C.POINTER(C.c_int32), _VP, _VP, _VP, _VP]),
    "ttx_debug_gemm_bench": (C.c_int, [_VP, _I, _I, _I, _I, _I, _I, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ttx_last_kernel_profile": (C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
}

_lib = None


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build() -> bool:
    if not LIB_PATH.exists():
        return True
    t = LIB_PATH.stat().st_mtime
    return any(src.stat().st_mtime > t for src in SOURCES)


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile the HIP library for gfx950 in-tree (cross-compiles without a GPU): one object per translation unit, in
    parallel, objects newer than their unit and every header are kept."""
    if not force and not needs_build():
        return LIB_PATH
    OBJ_DIR.mkdir(exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{INCLUDE}"] + os.environ.get("TTX_HIPCC_FLAGS", "").split()
    newest_header = max(h.stat().st_mtime for h in HEADERS)
    jobs, objs = [], []
    for unit in UNITS:
        obj = OBJ_DIR / (unit.stem + ".o")
        objs.append(str(obj))
        if not force and obj.exists() and obj.stat().st_mtime > max(unit.stat().st_mtime, newest_header) \
                and not os.environ.get("TTX_HIPCC_FLAGS"):
            continue
        cmd = [hipcc_path(), *flags, "-c", str(unit), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, proc in jobs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB_PATH), *objs]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


def lib():
    """Load (building first if the sources are newer) and type every exported symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if needs_build():
        build()
    handle = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(handle, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if handle.ttx_abi_version() != 4:
        raise RuntimeError("libttx_hip.so ABI version mismatch")
    _lib = handle
    return _lib


def check(code: int) -> None:
    if code == TTX_OK:
        return
    msg = lib().ttx_last_error().decode("utf-8", "replace")
    if code == TTX_ERR_REFERENCE:
        raise ReferenceError_(msg)
    if code == TTX_ERR_MAX_STEPS:
        raise RuntimeError("beam-speculative loop exceeded max_steps (non-terminating input)")
    raise TtxError(code, msg)
