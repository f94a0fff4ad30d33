"""ORACLE (test infrastructure): CPU restatement of the reference draft maker.

Follows src/utils/drafting.py:5-67 (``make_drafts``): every stride-1 window of length D over the
(right-padded) source is a candidate draft; N of them are picked at evenly spaced window indices,
the spacing being computed in *float32* (drafting.py:61-63) and truncated; EOS/PAD inside the
chosen windows are replaced by the replacement token (drafting.py:65-66).

Pinned by tests/golden/drafts.npz (outputs of the reference itself over the grid of the reference's
tests/test_drafting.py:19-21 and the call shapes of speculative_decoding.py:64-73, :430, :603-615).
"""
from __future__ import annotations

import numpy as np
import torch


def make_drafts(src, draft_len: int, n_drafts: int, min_draft_len: int, max_draft_len: int,
                eos_token_idx: int, pad_token_idx: int, replace_token_idx: int):
    """src: integer [B, L] (torch tensor or numpy array).  Returns the same kind, [B, N, D]."""
    # argument checks mirror drafting.py:39-43 (same failures for the same inputs)
    assert n_drafts > 0, "The number of drafts must be greater than 0"
    assert min_draft_len <= max_draft_len
    assert pad_token_idx != replace_token_idx
    assert eos_token_idx != replace_token_idx
    assert eos_token_idx != pad_token_idx

    as_torch = isinstance(src, torch.Tensor)
    s = src.detach().cpu().numpy() if as_torch else np.asarray(src)
    s = s.astype(np.int64, copy=True)
    B, L = s.shape
    N = n_drafts
    D = min(max(min_draft_len, draft_len), max_draft_len)          # drafting.py:48

    need = N + D - 1                                               # drafting.py:51-53
    if L < need:
        s = np.concatenate([s, np.full((B, need - L), pad_token_idx, dtype=np.int64)], axis=1)
    Lp = s.shape[1]
    W = Lp - D + 1                                                 # number of windows (drafting.py:57)

    service = (s == eos_token_idx) | (s == pad_token_idx)
    csum = np.concatenate([np.zeros((B, 1), dtype=np.int64), np.cumsum(service, axis=1)], axis=1)
    per_window = csum[:, D:D + W] - csum[:, :W]                    # service tokens inside each window
    n_clean = (per_window == 0).sum(axis=1)                        # drafting.py:58-60
    take_from = np.maximum(n_clean, N)                             # drafting.py:61

    # drafting.py:63 — int64 / python int is a float32 true-division in torch, the product with the
    # int64 step index is float32 too, then truncated by .long()
    ratio = (take_from - 1).astype(np.float32) / np.float32(max(N - 1, 1))
    index = (np.arange(N, dtype=np.int64).astype(np.float32)[None, :] * ratio[:, None]).astype(np.int64)

    gather = index[:, :, None] + np.arange(D, dtype=np.int64)[None, None, :]
    out = np.take_along_axis(s[:, None, :].repeat(N, axis=1), gather, axis=2)
    out[(out == eos_token_idx) | (out == pad_token_idx)] = replace_token_idx
    if as_torch:
        return torch.from_numpy(out).to(src.device)
    return out
