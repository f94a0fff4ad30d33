"""ORACLE (test infrastructure): CPU restatement of the reference's beam-search speculative generator.

  BeamSearchSpeculativeOracle <- TranslationInferenceBeamSearchSpeculative
                                 src/decoding/speculative_decoding.py:241-869
      generate_trying_all_the_drafts   :428-598        sample                       :294-400
      generate_with_smart_drafts       :600-845        get_vocab_tokens_bool_lib    :402-420
      calculate_n_accepted_in_drafts   :847-869
  nucleus_mask        <- mask_with_num_logits_according_nucleus   :871-904
  topk_per_group      <- topk_in_each_group                       :177-238

The floating-point parts (sort / softmax / cumsum / log, all fp32) are evaluated with the same torch
primitives in the same order as the reference, because candidate ranking compares fp32 sums.  The
candidate bookkeeping — which the reference expresses as boolean-mask scatter/gather over padded 2-D
tensors — is restated here with explicit per-candidate Python state (token lists), which is what makes
this file a readable specification of the algorithm rather than a transliteration.

Pinned by tests/golden/gen_spec_beam.npz and helpers.npz (outputs of the reference itself).
A ``max_steps`` guard is an ORACLE ADDITION: the reference loop does not terminate when a low-ranked
candidate keeps emitting PAD before any EOS (observed on the overfit tiny model; see
tests/golden/make_golden.py:section_spec_beam).
"""
from __future__ import annotations

import numpy as np
import torch

from .drafting import make_drafts

NEG_INF = float("-inf")


def nucleus_mask(logits: torch.Tensor, nucleus: float, max_kept: int, fill) -> torch.Tensor:
    """Per distribution keep the best logit and further ones, best first, while the probability mass of
    the logits ranked above stays below ``nucleus``; never more than ``max_kept``; the rest become
    ``fill`` (speculative_decoding.py:883-904)."""
    shape = logits.shape
    flat = logits.reshape(-1, shape[-1])
    srt, order = torch.sort(flat, descending=True)
    mass_above = torch.cumsum(srt.softmax(-1), dim=-1).roll(1, dims=-1)
    mass_above[:, 0] = nucleus - 1
    keep = mass_above < nucleus
    keep[:, max_kept:] = False
    srt = srt.masked_fill(~keep, float(fill))
    return torch.gather(srt, 1, order.argsort(1)).reshape(shape)


def topk_per_group(score: torch.Tensor, lengths, k: int, pad=None):
    """Largest k scores of every consecutive group (group g owns ``lengths[g]`` entries); returns
    (scores [G,k], flat indices [G*k]) best first (speculative_decoding.py:177-238)."""
    lengths = [int(x) for x in lengths]
    assert min(lengths) >= k
    flat = score.reshape(-1)
    L = max(lengths)
    starts = np.concatenate([[0], np.cumsum(lengths)[:-1]])
    if pad is None:
        pad = flat.min().item() - 1
    table = torch.full((len(lengths), L), pad, dtype=flat.dtype)
    for g, (s, n) in enumerate(zip(starts, lengths)):
        table[g, :n] = flat[s:s + n]
    top, idx = table.topk(k, dim=-1, sorted=True)
    return top, (idx + torch.as_tensor(starts).unsqueeze(1)).reshape(-1)


class BeamSearchSpeculativeOracle:
    def __init__(self, model, max_len: int, n_best: int, draft_len: int, n_drafts: int, vocab_size: int,
                 smart_drafts_mode: bool, pad_token: int, bos_token: int, eos_token: int, C_token: int,
                 max_steps: int | None = None) -> None:
        self.model = model
        self.max_len = max_len
        self.vocab_size = vocab_size
        self.smart_drafts_mode = smart_drafts_mode
        self.pad, self.bos, self.eos, self.c_tok = pad_token, bos_token, eos_token, C_token
        self.n_best = n_best
        self.requested_drafts_num = n_drafts
        self.min_draft_len, self.max_draft_len = 5, 200                   # :278-284
        self.draft_len = min(max(self.min_draft_len, draft_len), self.max_draft_len)
        self.accepted_tokens_num = 0
        self.model_calls_num = 0
        self.model_input_lines_num = 0
        self.produced_non_pad_tokens = 0
        self.max_steps = max_steps
        # tests/test_beam_pool_replay.py: a list here receives one record per iteration with, per source, what the HIP source
        # pool traces (longest new row, draft-group data, counter contributions)
        self.trace = None

    def __str__(self):
        return (f"SpeculativeSampling decoding (n_best={self.n_best}, max_len={self.max_len}, "
                f"max_num_of_drafts={self.requested_drafts_num}, draft_len={self.draft_len})")

    def generate(self, src: torch.Tensor) -> torch.Tensor:
        return self._run(src, smart=self.smart_drafts_mode)

    # ---------------------------------------------------------------------------------------------
    def _draft_slots(self, row: np.ndarray, dl: int) -> np.ndarray:
        """Column indices of the first ``dl`` PAD entries of a candidate row (:497-500)."""
        return np.flatnonzero(row == self.pad)[:dl]

    def _verify_positions(self, slots: np.ndarray, width: int) -> np.ndarray:
        """The dl+1 columns whose logits are examined: every draft slot and the column before it
        (``draft_place | roll(draft_place, -1)``, :524-526).  roll wraps around the row end."""
        marks = np.zeros(width, dtype=bool)
        marks[slots] = True
        marks |= np.roll(marks, -1)
        return np.flatnonzero(marks)

    def _run(self, src: torch.Tensor, smart: bool) -> torch.Tensor:
        K, V, PAD, EOS = self.n_best, self.vocab_size, self.pad, self.eos
        B = src.size(0)
        mask = src == self.model.src_pad_token_i
        memory = self.model.encode_src(src, mask)
        src_np = src.cpu().numpy()
        if smart:
            # library of every window, BOS included; only windows whose first token equals the candidate's
            # last token are tried, at most `requested` of them (:603-618, :402-420)
            lib = make_drafts(src_np, self.draft_len + 1, src_np.shape[1] - 5, self.min_draft_len, self.max_draft_len,
                              EOS, PAD, self.c_tok)
            dl = lib.shape[2] - 1
        else:
            drafts_all = make_drafts(src_np[:, 1:], self.draft_len, self.requested_drafts_num, self.min_draft_len,
                                     self.max_draft_len, EOS, PAD, self.c_tok)          # [B,N,D] (:430-431)
            dl = drafts_all.shape[2]

        cands = [np.array([self.bos], dtype=np.int64) for _ in range(B)]       # candidate rows, all same width
        owner = list(range(B))                                                 # source index of each candidate
        logp = torch.zeros(B, dtype=torch.float32)
        empty_cols = 0
        after_last = 1
        room = self.max_len - after_last - 1
        steps = 0
        result = None
        while room >= 1 and after_last <= self.max_len:                        # :464 / :652
            if self.max_steps is not None and steps >= self.max_steps:
                raise RuntimeError("beam-speculative loop exceeded max_steps (non-terminating input)")
            steps += 1
            dl = min(room, dl)
            grow = dl + 1 - empty_cols
            if grow > 0:
                cands = [np.concatenate([c, np.full(grow, PAD, dtype=np.int64)]) for c in cands]
            width = len(cands[0])
            n_cand = len(cands)

            # ---- decoder rows: (candidate, draft) pairs
            rows_cand, rows_draft = [], []
            for ci, row in enumerate(cands):
                b = owner[ci]
                if smart:
                    last = row[int((row != PAD).sum()) - 1]                    # :690-695
                    match = np.flatnonzero(lib[b, :, 0] == last)
                    if match.size == 0:
                        match = np.array([0])                                   # "each line needs at least one draft"
                    for j in match[:self.requested_drafts_num]:
                        rows_cand.append(ci)
                        rows_draft.append(lib[b, j, 1:dl + 1])
                else:
                    for n in range(drafts_all.shape[1]):
                        rows_cand.append(ci)
                        rows_draft.append(drafts_all[b, n, :dl])
            slots = [self._draft_slots(row, dl) for row in cands]
            inputs = np.stack([cands[ci].copy() for ci in rows_cand])
            for r, (ci, d) in enumerate(zip(rows_cand, rows_draft)):
                inputs[r, slots[ci]] = d
            self.model_calls_num += 1
            if smart:                                                          # only the smart-drafts loop counts them (:741)
                self.model_input_lines_num += len(rows_cand)
            running = ~(inputs == EOS).any(axis=1)
            logits = torch.zeros((len(rows_cand), dl + 1, V), dtype=torch.float32)
            logits[:, :, PAD] = 35.0                                           # finished rows: ~certain PAD (:466-468)
            if running.any():
                sel = np.flatnonzero(running)
                mem_rows = torch.as_tensor([owner[rows_cand[r]] for r in sel])
                out = self.model.decode_tgt(torch.from_numpy(inputs[sel]).to(src.device), memory[mem_rows],
                                            memory_pad_mask=mask[mem_rows]).cpu()
                for i, r in enumerate(sel):
                    cols = self._verify_positions(slots[rows_cand[r]], width)
                    logits[r] = out[i, cols, :]

            # ---- accepted length of every draft: leading draft tokens that survive the 0.9975 nucleus (:539-548)
            probs = nucleus_mask(logits, 0.9975, K, "-inf").softmax(-1)
            draft_t = torch.from_numpy(np.stack(rows_draft))
            alive = torch.gather(probs[:, :-1, :], 2, draft_t.unsqueeze(-1)).squeeze(-1) != 0.0
            n_ok = torch.cumprod(alive.long(), dim=1).sum(dim=1)                # [rows]

            # ---- best draft per candidate (:553-565 / :779-789)
            per_cand = np.bincount(rows_cand, minlength=n_cand)
            if smart:
                best_n, best_row = topk_per_group(n_ok, per_cand, 1, pad=-1)
                best_n = best_n.reshape(-1)
            else:
                N = drafts_all.shape[1]
                best_n, which = n_ok.reshape(n_cand, N).topk(1, dim=-1)
                best_n = best_n.reshape(-1)
                best_row = torch.arange(n_cand) * N + which.reshape(-1)
            rec = None
            if self.trace is not None:
                beam_now = 1 if n_cand == B else K
                fin_c = [bool((cands[ci] == EOS).any()) for ci in range(n_cand)]
                rec = {"lines": np.zeros(B, int), "running": np.zeros(B, int), "grp": np.zeros(B, int), "sens": np.zeros(B, bool),
                       "run_cands": np.zeros(B, int)}
                starts = np.concatenate([[0], np.cumsum(per_cand)[:-1]])
                for ci in range(n_cand):
                    b = ci // beam_now
                    rec["lines"][b] += per_cand[ci]
                    rec["grp"][b] = max(rec["grp"][b], per_cand[ci])
                    if not fin_c[ci]:
                        rec["running"][b] += per_cand[ci]
                        rec["run_cands"][b] += 1
                if smart:
                    for ci in range(n_cand):
                        b = ci // beam_now
                        if fin_c[ci]:
                            continue
                        vals = n_ok[starts[ci]:starts[ci] + per_cand[ci]]
                        def pick(width):
                            row = torch.full((width,), -1, dtype=vals.dtype)
                            row[:len(vals)] = vals
                            return int(row.topk(1).indices[0])
                        own = pick(int(rec["grp"][b]))
                        if any(pick(w) != own for w in range(int(rec["grp"][b]) + 1, self.requested_drafts_num + 1)):
                            rec["sens"][b] = True
            chosen = draft_t[best_row].clone()                                  # [n_cand, dl]
            cl = logits[best_row]                                               # [n_cand, dl+1, V]

            # ---- every single-token deviation along the accepted prefix becomes a leaf (:320-400)
            tree = nucleus_mask(cl, 20.0, K, 0.0)                               # top-K logits kept, rest 0
            pos = torch.arange(dl + 1)
            tree = tree * (pos.unsqueeze(0) <= best_n.unsqueeze(1)).unsqueeze(-1)
            short = best_n != dl
            chosen[short, best_n[short]] = self.bos                            # :338-339
            tree[:, :-1, :].scatter_(2, chosen.unsqueeze(-1), 0.0)              # accepted tokens cannot be leaves
            leaf_c, leaf_p, leaf_t = torch.nonzero(tree, as_tuple=True)
            lp = cl.softmax(-1).log()
            beam = 1 if n_cand == B else K
            assert n_cand == B * beam
            per_src = np.bincount(np.asarray(leaf_c) // beam, minlength=B)

            new_rows, new_scores, acc_mark = [], [], []
            for c, p, t in zip(leaf_c.tolist(), leaf_p.tolist(), leaf_t.tolist()):
                seq = torch.cat([chosen[c], torch.zeros(1, dtype=torch.long)])
                seq[p] = t
                step_lp = torch.gather(lp[c], 1, seq.unsqueeze(-1)).squeeze(-1)
                step_lp = step_lp.masked_fill(pos > p, 0.0).cumsum(-1)          # fp32 running sum, as :382-384
                new_scores.append(logp[c] + step_lp[-1])
                seq = seq.masked_fill(pos > p, PAD).numpy()
                row = cands[c].copy()
                place = np.zeros(width, dtype=bool)
                place[slots[c]] = True
                place |= np.roll(place, 1)                                      # :393
                row[np.flatnonzero(place)] = seq
                new_rows.append(row)
                acc_mark.append(-1 if (cands[c] == EOS).any() else p)           # :397

            scores = torch.stack(new_scores)
            top_s, top_i = topk_per_group(scores, per_src, K, pad=NEG_INF)
            top_i = top_i.tolist()
            cands = [new_rows[i] for i in top_i]
            owner = [b for b in range(B) for _ in range(K)]
            kept = [acc_mark[i] for i in top_i if acc_mark[i] >= 0]
            self.accepted_tokens_num += int(sum(kept))
            self.produced_non_pad_tokens += int(sum(kept)) + len(kept)
            result = np.stack(cands)
            if rec is not None:
                rec["longest"] = np.array([max(int((row != PAD).sum()) for row in cands[b * K:(b + 1) * K]) for b in range(B)])
                rec["n_eos"] = np.array([sum(bool((row == EOS).any()) for row in cands[b * K:(b + 1) * K]) for b in range(B)])
                rec["acc_sum"] = np.zeros(B, int)
                rec["acc_cnt"] = np.zeros(B, int)
                for r_, i in enumerate(top_i):
                    if acc_mark[i] >= 0:
                        rec["acc_sum"][r_ // K] += acc_mark[i]
                        rec["acc_cnt"][r_ // K] += 1
                self.trace.append(rec)
            if all((row == EOS).any() for row in cands):                        # :586 / :826-829
                break
            logp = top_s.reshape(-1)
            empty_cols = int(min((row == PAD).sum() for row in cands))
            after_last = width - empty_cols
            room = self.max_len - after_last - 1
        return torch.from_numpy(result.reshape(B, K, -1)).to(src.device)
