"""Generators with the reference's constructor keywords, counters and ``generate`` contract
(src/model/lightning_model.py:92-137 shows how they are built), running on the
HIP library.  ``model`` must be a ``NativeTransformer``.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _native as N
from .model import NativeTransformer


def _need_native(model) -> NativeTransformer:
    if not isinstance(model, NativeTransformer):
        raise TypeError("the native generators run on a NativeTransformer (no eager/PyTorch fallback exists)")
    return model


class TranslationInferenceGreedySpeculative:
    """Drop-in for src/decoding/speculative_decoding.py:8-174: copy-drafts, one parallel verify pass per
    step, longest accepted prefix + 1 bonus token — the whole loop runs in ttx_greedy_speculative_generate."""

    def __init__(self, model, max_len: int, draft_len: int, n_drafts: int, pad_token: int, bos_token: int,
                 eos_token: int, replace_token: int) -> None:
        self.model = _need_native(model)
        self.max_len = max_len
        self.pad_token, self.bos_token, self.eos_token = pad_token, bos_token, eos_token
        self.replace_token = replace_token
        self.draft_len = draft_len
        self.n_drafts = n_drafts
        self.accepted_tokens_num = 0   # left at 0 by the reference's greedy-speculative loop as well
        self.model_calls_num = 0
        self.stats_total = {"accepted_tokens": 0, "produced_tokens": 0, "verified_positions": 0,
                            "kv_prefix_positions": 0, "src_positions": 0, "encode_ms": 0.0, "decode_ms": 0.0,
                            "src_tokens_padded": 0, "batches": 0}
        self.last_stats: N.GenStats | None = None

    def __str__(self):
        return (f"Greedy speculative decoding (draft_len={self.draft_len}, n_drafts={self.n_drafts}, "
                f"max_len={self.max_len})")

    def generate(self, src: torch.Tensor) -> torch.Tensor:
        m = self.model
        src = src.to(m.device, torch.int64).contiguous()
        B, Ls = src.shape
        out = torch.empty((B, 1, self.max_len), dtype=torch.int64, device=m.device)
        p = N.GenParams(self.max_len, self.draft_len, self.n_drafts, self.pad_token, self.bos_token, self.eos_token,
                        self.replace_token, 0)
        st = N.GenStats()
        N.check(m._lib.ttx_greedy_speculative_generate(m.session, src.data_ptr(), B, Ls, C.byref(p), out.data_ptr(),
                                                       C.byref(st), m._stream()))
        self.model_calls_num += int(st.model_calls)
        t = self.stats_total
        for k in ("accepted_tokens", "produced_tokens", "verified_positions", "kv_prefix_positions", "src_positions",
                  "encode_ms", "decode_ms"):
            t[k] += getattr(st, k)
        t["src_tokens_padded"] += B * Ls
        t["batches"] += 1
        self.last_stats = st
        return out
