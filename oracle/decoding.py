"""ORACLE (test infrastructure): CPU restatement of the reference's greedy, beam-search and
greedy-speculative generators.  (The beam-speculative generator lives in oracle/spec_beam.py.)

  * GreedyOracle            <- TranslationInferenceGreedy            src/decoding/standard_decoding.py:4-55
  * BeamSearchOracle        <- TranslationInferenceBeamSearch        src/decoding/standard_decoding.py:58-174
  * GreedySpeculativeOracle <- TranslationInferenceGreedySpeculative src/decoding/speculative_decoding.py:8-174

Each class keeps the reference's constructor keywords, counters and ``generate(src) -> Long[B,N,L]``
contract and drives any object with the model protocol B5 (``src_pad_token_i``,
``encode_src``, ``decode_tgt``, ``__call__``).  Like the reference they re-decode the whole prefix on
every step (no KV cache) — this is what bench.py times as ``cpu_baseline`` (kind "port").

Token bookkeeping is done on the host in numpy with explicit per-row state instead of the reference's
chains of tensor scatter/gather calls; the decoder is called with exactly the same integer inputs.

Pinned by tests/golden/gen_greedy.npz, gen_beam.npz, gen_spec_greedy.npz (outputs of the reference
itself on the tiny trained model; tests/test_oracle_decoding.py).
"""
from __future__ import annotations

import numpy as np
import torch

from .drafting import make_drafts


def _np(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().numpy()


class GreedyOracle:
    def __init__(self, model, max_len: int, pad_token: int, bos_token: int, eos_token: int) -> None:
        self.model = model
        self.max_len = max_len
        self.pad_token, self.bos_token, self.eos_token = pad_token, bos_token, eos_token
        self.model_calls_num = 0
        self.given_tokens = 0

    def __str__(self):
        return f"Greedy decoding (max_len={self.max_len})"

    def generate(self, src: torch.Tensor) -> torch.Tensor:
        B = src.size(0)
        out = torch.full((B, self.max_len), self.pad_token, dtype=src.dtype, device=src.device)
        out[:, 0] = self.bos_token
        mask = src == self.model.src_pad_token_i
        self.given_tokens += int((~mask).sum())
        memory = self.model.encode_src(src, mask)
        for i in range(1, self.max_len):                                  # standard_decoding.py:45
            logits = self.model.decode_tgt(out[:, :i], memory, memory_pad_mask=mask)
            self.model_calls_num += 1
            nxt = logits[:, -1, :].argmax(dim=-1)
            out[:, i] = nxt
            # stop only when EVERY row emitted EOS or PAD at this very step (standard_decoding.py:51-53)
            if bool(((nxt == self.eos_token) | (nxt == self.pad_token)).all()):
                break
        return out.unsqueeze(1)


class BeamSearchOracle:
    def __init__(self, model, beam_size: int, max_len: int, pad_token: int, bos_token: int, eos_token: int):
        assert max_len > 1 and beam_size > 0
        self.model = model
        self.beam_size = beam_size
        self.max_len = max_len
        self.pad_token, self.bos_token, self.eos_token = pad_token, bos_token, eos_token
        self.model_calls_num = 0
        self.given_tokens = 0
        self.b_sz = 0

    def __str__(self):
        return f"Beam search decoding (beam_size={self.beam_size}, max_len={self.max_len})"

    def generate(self, src: torch.Tensor) -> torch.Tensor:
        K = self.beam_size
        B = src.size(0)
        pad_col = self.model.src_pad_token_i                          # standard_decoding.py:135 (quirk 6)
        dev = src.device

        # step 0: one full forward on <BOS> (standard_decoding.py:99-107)
        y0 = torch.full((B, 1), self.bos_token, dtype=src.dtype, device=dev)
        first = self.model(src, y0)
        self.b_sz += B
        self.model_calls_num += 1
        self.given_tokens += int((src != pad_col).sum())
        logp = torch.log(torch.softmax(first, dim=-1))                # log(softmax), NOT log_softmax
        V = logp.size(-1)
        score, tok = torch.topk(logp[:, 0, :], K, dim=-1, sorted=True)   # [B,K]
        y = torch.cat([torch.full((B * K, 1), self.bos_token, dtype=src.dtype, device=dev),
                       tok.reshape(-1, 1)], dim=1)

        # the source is re-encoded once per beam (standard_decoding.py:120-124)
        src_rep = src.repeat_interleave(K, dim=0)
        mask_rep = src_rep == pad_col
        memory = self.model.encode_src(src_rep, mask_rep)

        for _ in range(self.max_len - 2):                              # standard_decoding.py:126,130
            alive = ~((y == self.eos_token).any(dim=1))
            self.b_sz += int(alive.sum())
            step_logits = torch.zeros((B * K, V), dtype=torch.float32, device=dev)
            step_logits[:, pad_col] = 35.0                              # finished rows: ~certain PAD
            live_logits = self.model.decode_tgt(y[alive], memory[alive], memory_pad_mask=mask_rep[alive])
            step_logits[alive] = live_logits[:, -1, :]
            self.model_calls_num += 1
            nxt = torch.log(torch.softmax(step_logits, dim=-1)).reshape(B, K, V)
            total = (score.unsqueeze(-1) + nxt).reshape(B, K * V)
            score, flat = total.topk(K, dim=-1, sorted=True)
            parent = torch.div(flat, V, rounding_mode="floor") + torch.arange(B, device=dev).unsqueeze(1) * K
            y = torch.cat([y[parent.reshape(-1)], (flat % V).reshape(-1, 1)], dim=1)
            if bool((y == self.eos_token).any(dim=1).all()):
                break
        return y.reshape(B, K, -1)


class GreedySpeculativeOracle:
    def __init__(self, model, max_len: int, draft_len: int, n_drafts: int, pad_token: int, bos_token: int,
                 eos_token: int, replace_token: int) -> None:
        self.model = model
        self.max_len = max_len
        self.pad_token, self.bos_token, self.eos_token = pad_token, bos_token, eos_token
        self.replace_token = replace_token
        self.draft_len = draft_len
        self.n_drafts = n_drafts
        self.accepted_tokens_num = 0      # never updated by the reference's greedy-speculative loop either
        self.model_calls_num = 0
        # extra, oracle-only: totals and one record per verify step (roofline accounting of bench.py)
        self.accepted_total = 0
        self.step_log: list[tuple[int, int, int]] = []
        # oracle-only: (running rows, their fronts) after every verify step of the last generate call —
        # the per-row traces tests/test_row_scheduling.py replays
        self.front_log: list[tuple[np.ndarray, np.ndarray]] = []

    def __str__(self):
        return (f"Greedy speculative decoding (draft_len={self.draft_len}, n_drafts={self.n_drafts}, "
                f"max_len={self.max_len})")

    def generate(self, src: torch.Tensor) -> torch.Tensor:
        N, D = self.n_drafts, self.draft_len
        PAD, EOS = self.pad_token, self.eos_token
        B = src.size(0)
        dev = src.device
        mask = src == self.model.src_pad_token_i
        memory = self.model.encode_src(src, mask)                       # once per batch (:61)
        drafts = _np(make_drafts(src[:, 1:], D, N, 1, self.max_len, EOS, PAD, self.replace_token))  # [B,N,D]
        assert drafts.shape[1] == N
        Dd = drafts.shape[2]                                            # == D unless clamped by max_len

        result = np.full((B, self.max_len), PAD, dtype=np.int64)
        self.front_log = []
        alive = np.arange(B)                                            # original indices of running rows
        gen = np.full((B, 1), self.bos_token, dtype=np.int64)           # [Bc, Lg]
        front = np.zeros(B, dtype=np.int64)                             # index of the last real token

        while gen.shape[1] < self.max_len:                              # :93
            Bc = len(alive)
            # columns that are PAD in every running row are dropped before D+1 fresh ones are added (:97-102)
            empty_cols = int(((gen == PAD).sum(axis=0) == Bc).sum())
            grow = Dd + 1 - empty_cols
            if grow >= 0:
                gen = np.concatenate([gen, np.full((Bc, grow), PAD, dtype=np.int64)], axis=1)
            else:  # F.pad with a negative amount trims: happens after long rows have retired
                gen = gen[:, :grow]
            width = gen.shape[1]

            # decoder input: every running row repeated N times, draft n written after its front (:104-115)
            inp = np.repeat(gen[:, :-1], N, axis=0)                      # [Bc*N, width-1]
            cols = front.repeat(N)[:, None] + 1 + np.arange(Dd)[None, :]
            if cols.max(initial=0) >= inp.shape[1]:
                raise RuntimeError("index out of range in scatter (reference quirk 2)")
            np.put_along_axis(inp, cols, drafts[alive].reshape(Bc * N, Dd), axis=1)

            rows = np.repeat(alive, N)
            logits = self.model.decode_tgt(torch.from_numpy(inp).to(dev), memory[rows], memory_pad_mask=mask[rows])
            self.model_calls_num += 1
            self.step_log.append((Bc, int(front.max()) + 1, int(src.size(1))))
            pred = _np(logits.argmax(dim=2))                            # [Bc*N, width-1]
            look = front.repeat(N)[:, None] + np.arange(Dd + 1)[None, :]
            pred = np.take_along_axis(pred, look, axis=1).reshape(Bc, N, Dd + 1)

            # longest verified prefix per draft, best draft per row (:129-137)
            agree = drafts[alive] == pred[:, :, :-1]                     # [Bc,N,D]
            n_ok = np.cumprod(agree, axis=2).sum(axis=2)                 # leading run of matches
            best = n_ok.argmax(axis=1)                                   # first maximum; ties are harmless (quirk 5)
            n_acc = n_ok[np.arange(Bc), best]
            chosen = pred[np.arange(Bc), best]                           # [Bc, D+1]
            keep = np.arange(Dd + 1)[None, :] <= n_acc[:, None]
            chosen = np.where(keep, chosen, PAD)
            np.put_along_axis(gen, front[:, None] + 1 + np.arange(Dd + 1)[None, :], chosen, axis=1)  # (:144-145)
            front = front + n_acc + 1
            self.accepted_total += int(n_acc.sum())
            self.front_log.append((alive.copy(), front.copy()))

            done = (gen == EOS).any(axis=1)                              # (:149-168)
            if done.any():
                if width > self.max_len:
                    raise RuntimeError("shape mismatch writing a finished row wider than max_len")
                result[alive[done], :width] = gen[done]
                alive, gen, front = alive[~done], gen[~done], front[~done]
            if len(alive) == 0:
                break
        return torch.from_numpy(result).to(dev).unsqueeze(1)
