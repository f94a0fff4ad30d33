"""Pure-torch make_drafts used ONLY when the beam host logic is unit-tested on CPU with a non-native model
(tests/test_host_beam_logic.py).  The product path (NativeTransformer) always calls ttx_make_drafts; this
module is never reached with the HIP library in use."""
from __future__ import annotations

import torch


def make_drafts(src: torch.Tensor, draft_len: int, n_drafts: int, lo: int, hi: int, eos: int, pad: int, repl: int):
    assert n_drafts > 0 and lo <= hi and len({eos, pad, repl}) == 3
    B, L = src.shape
    N, D = n_drafts, min(max(lo, draft_len), hi)
    need = N + D - 1
    s = src if L >= need else torch.cat([src, src.new_full((B, need - L), pad)], dim=1)
    win = s.unfold(1, D, 1)
    clean = ((win == eos) | (win == pad)).sum(-1).eq(0).sum(-1)
    take = torch.clamp(clean, min=N).view(B, 1)
    idx = (torch.arange(N, device=s.device) * ((take - 1) / max(N - 1, 1))).long()
    out = win.gather(1, idx.unsqueeze(-1).expand(-1, -1, D)).clone()
    out[(out == eos) | (out == pad)] = repl
    return out
