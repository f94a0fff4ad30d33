"""Row scheduling for greedy-speculative decoding: decode rows in length-sorted groups, then replay the
reference's per-batch loop over the batches the caller actually passed.

Why this is exact.  In `TranslationInferenceGreedySpeculative.generate`
(src/decoding/speculative_decoding.py:39-174) the rows of a batch interact in one way only: they share the
tensor `generated_tokens`, whose width after an iteration is  max(front of the running rows) + draft_len + 2
(:97-102 drop the all-PAD columns, then add draft_len + 1).  That width
  * ends the loop once it reaches max_len (:93) — rows still running stay all-PAD in the result, and
  * makes the write of a finished row raise when it exceeds max_len (:158).
Tokens, drafts and accepted lengths of a row depend on that row alone (drafts: src/utils/drafting.py:5-67;
the HIP path's arithmetic is batch-invariant).  So if every row is decoded under the rule it
would see alone in a batch and its front after every step is kept (`ttx_greedy_speculative_generate_rows`),
the behaviour of ANY grouping of those rows follows from the traces by integer bookkeeping — done here in
numpy, on the host, per original batch.

The one exception is reference quirk 2 (a PAD token inside a sequence changes the all-PAD column count);
the library reports it (TTX_ERR_ROW_REPLAY) and the caller decodes the batches as given.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class BatchReplay:
    model_calls: int          # iterations the reference's loop makes on this batch (model_calls_num increment)
    error: bool               # the reference raises on this batch (:158, width beyond max_len)
    finished: np.ndarray      # bool [B]: rows whose tokens the reference returns (others stay all-PAD)
    accepted_tokens: int      # draft tokens accepted over the iterations that ran
    produced_tokens: int      # accepted + one bonus token per row and iteration
    verified_positions: int   # decoder positions computed: rows * (1 + n_drafts * draft_len) per iteration
    kv_prefix_positions: int  # cached prefix positions attended
    rows_iterations: int      # sum over iterations of running rows


def replay_batch(traj: np.ndarray, fin_step: np.ndarray, max_len: int, draft_len: int, n_drafts: int = 1) -> BatchReplay:
    """traj: int [B, max_len + 1], traj[r, t] = front of row r after its t-th verify step (column 0 is 0, -1
    past the row's last step); fin_step: int [B], step at which row r produced EOS (0: it never did).

    The reference's loop, all iterations at once: row r takes part in iteration t while it has not finished
    (fin_step[r] == 0 or >= t); the width after iteration t is max(front before t over those rows) + draft_len + 2;
    iteration t happens iff every earlier one did, somebody is still running and the width before it is < max_len."""
    traj = np.asarray(traj).astype(np.int64)
    fin = np.asarray(fin_step).astype(np.int64)
    B, T = traj.shape[0], traj.shape[1] - 1
    t = np.arange(1, T + 1)
    takes_part = (fin == 0)[:, None] | (fin[:, None] >= t[None, :])          # [B, T]
    before, after = traj[:, :-1], traj[:, 1:]
    width = np.where(takes_part, before, -1).max(axis=0) + draft_len + 2     # after iteration t (:97-102, :145)
    width_before = np.concatenate([[1], width[:-1]])
    happens = np.logical_and.accumulate(takes_part.any(axis=0) & (width_before < max_len))   # :93
    finishing = (fin[:, None] == t[None, :]).any(axis=0)
    raises = happens & finishing & (width > max_len)                         # :158 — shape mismatch in the reference
    error = bool(raises.any())
    if error:
        happens = happens & (t <= t[np.argmax(raises)])                      # the loop dies inside that iteration
    n_it = int(happens.sum())
    m = takes_part & happens[None, :]
    if ((before < 0) | (after < 0))[m].any():
        raise ValueError("row trace shorter than the batch needs: traces were not produced under the per-row rule")
    finished = (fin >= 1) & (fin <= n_it) & (not error)
    rows_it = int(m.sum())
    adv = int((after - before)[m].sum())
    rps = 1 + n_drafts * draft_len
    return BatchReplay(n_it, error, finished, adv - rows_it, adv, rows_it * rps, int(before[m].sum()), rows_it)


def plan_row_groups(lengths, group_size: int):
    """Order rows by (unpadded) source length, longest first, and cut the order into groups of `group_size`.
    Returns (order, [slice, ...]); longest-first so the biggest workspaces are allocated once, up front."""
    lengths = np.asarray(lengths)
    order = np.argsort(-lengths, kind="stable")
    groups = [slice(i, min(i + group_size, len(order))) for i in range(0, len(order), group_size)]
    return order, groups
