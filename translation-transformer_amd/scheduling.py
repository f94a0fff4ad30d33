"""Row / source scheduling: decode rows (greedy-speculative) or sources (beam-speculative) regrouped on the device,
then replay the reference's per-batch loop over the batches the caller actually passed.

Part 1 — greedy-speculative rows.

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


# ---------------------------------------------------------------------------------------------------------------------
# Part 2 — beam-speculative batches (ttx_beam_speculative_generate_pool).
#
# The pool decodes the sources of a given batch in lock-step and keeps the batch-wide scalars of
# `TranslationInferenceBeamSearchSpeculative.generate_*` (src/decoding/speculative_decoding.py:428-598, :600-845) on the device:
# the draft length min(max_len - longest row - 1, draft_len) (:476 / :671), the stop rule (all rows hold EOS, :586 / :826; no
# room left, :464 / :652) and, in smart mode, the longest draft group (:779-784).  What is left for the host is what the
# reference derives from the shared TENSOR of the batch: its width, which grows to (longest row) + draft length + 1 before
# every iteration (:488-491 / :679-686), the number of iterations, and the counters — a source that finished (and gave its
# slots back) is a fixed point of the loop and keeps contributing n_best one-draft rows per iteration to
# `model_input_lines_num` (smart mode, :741) until its batch ends.

BP_DONE, BP_ERR_LEAVES, BP_STOPPED, BP_MAX_STEPS = 1, 2, 3, 4      # BeamPoolStatus (csrc/ttx_loop_kernels.hip.h)


@dataclass
class BeamBatchReplay:
    error: str | None         # None, "reference" (fewer leaves than n_best for a source: the reference asserts, :195) or "max_steps"
    model_calls: int = 0      # iterations of the reference's loop on this batch
    out_width: int = 0        # columns of the tensor the reference returns
    accepted_tokens: int = 0
    produced_non_pad_tokens: int = 0
    input_lines: int = 0      # smart mode: (candidate, draft) rows built over all iterations (model_input_lines_num)
    running_rows: int = 0     # smart mode: of those, rows of unfinished candidates (b_sz)


def replay_beam_batch(trace_len: np.ndarray, summary: np.ndarray, max_len: int, draft_len: int, n_best: int) -> BeamBatchReplay:
    """trace_len int [B, T_cap]: longest hypothesis of source b after its t-th iteration (column t - 1); summary int [B, 8]:
    iterations, status, input lines, running rows, accepted sum, accepted count, ... per source; draft_len: the loop's initial
    draft length (the clamped one; smart mode: tokens after the key token)."""
    summary = np.asarray(summary).astype(np.int64)
    B = summary.shape[0]
    T_s, status = summary[:, 0], summary[:, 1]
    if (status == BP_ERR_LEAVES).any():
        return BeamBatchReplay("reference")
    if (status == BP_MAX_STEPS).any():
        return BeamBatchReplay("max_steps")
    if not np.isin(status, (BP_DONE, BP_STOPPED)).all():
        raise ValueError("a source of this batch was never retired by the pool")
    T = int(T_s.max())
    tl = np.asarray(trace_len).astype(np.int64)[:, :T]
    alive = np.arange(T)[None, :] < T_s[:, None]                               # source b took part in iteration t + 1
    last = tl[np.arange(B), T_s - 1][:, None]
    longest = np.where(alive, tl, last)                                        # a finished source keeps its rows
    if (longest < 1).any():
        raise ValueError("source trace shorter than its iteration count")
    batch_longest = longest.max(axis=0)                                        # after iteration t + 1
    width, prev, dl = 1, 1, draft_len
    for t in range(T):
        dl = min(max_len - prev - 1, dl)                                        # :476 / :671
        width = max(width, prev + dl + 1)                                       # :488-491: width += max(0, dl + 1 - empty columns)
        prev = int(batch_longest[t])
    acc = int(summary[:, 4].sum())
    return BeamBatchReplay(None, T, width, acc, acc + int(summary[:, 5].sum()),
                           int(summary[:, 2].sum() + ((T - T_s) * n_best).sum()), int(summary[:, 3].sum()))
