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
# Part 2 — beam-speculative sources (ttx_beam_speculative_generate_pool).
#
# In `TranslationInferenceBeamSearchSpeculative.generate_*` (src/decoding/speculative_decoding.py:428-598, :600-845) the
# sources of a batch meet in four batch-wide scalars only:
#   * the tensor width: it grows to (longest row) + draft_len + 1 before every iteration (:488-491 / :679-686);
#   * the draft length min(max_len - longest row - 1, draft_len) (:476 / :671) — equal to draft_len for every source as long
#     as no row of the batch is within draft_len + 1 of max_len;
#   * the stop rule: all b_size * n_best rows hold EOS (:586 / :826), or no room is left (:464 / :652);
#   * smart mode: the best draft of a candidate is topk(1) over its accepted lengths padded with -1 to the LONGEST draft group
#     of the batch (:779-784 -> topk_in_each_group :225) — torch's CPU top-k breaks ties differently for different widths.
# Candidates, leaves, log-probs and the per-source top-n_best selection of a source depend on that source alone, and a source
# all of whose rows hold EOS is a fixed point of the iteration (one PAD leaf of log-prob +0 per row, same order).  So each
# source is decoded as if alone, with per-iteration traces (longest new row; longest draft group and whether a wider table
# would change a choice), and this function derives what the reference does with the batch AS GIVEN — or says that the
# batch's scalars would have coupled its sources, in which case the caller decodes that batch as given.

BP_DONE, BP_ERR_LEAVES, BP_IRREGULAR, BP_MAX_STEPS, BP_RUNAWAY = 1, 2, 3, 4, 5      # BeamPoolStatus (csrc/ttx_loop_kernels.hip.h)


@dataclass
class BeamBatchReplay:
    as_given: bool            # True: decode this batch with the per-batch entry point (coupled scalars, or an error to raise)
    model_calls: int = 0      # iterations of the reference's loop on this batch
    out_width: int = 0        # columns of the tensor the reference returns
    accepted_tokens: int = 0
    produced_non_pad_tokens: int = 0
    input_lines: int = 0      # smart mode: (candidate, draft) rows built over all iterations (model_input_lines_num)
    running_rows: int = 0     # smart mode: of those, rows of unfinished candidates (b_sz)


def replay_beam_batch(trace_len: np.ndarray, trace_grp: np.ndarray, summary: np.ndarray, max_len: int, draft_len: int,
                      n_best: int, smart: bool) -> BeamBatchReplay:
    """trace_len int [B, T_cap]: longest hypothesis of source b after its t-th iteration (column t - 1); trace_grp uint8
    [B, T_cap]: bits 0-6 the source's longest draft group in that iteration, bit 7 "a wider table picks another draft";
    summary int [B, 8]: iterations, status, input lines, running rows, accepted sum, accepted count, ... per source;
    draft_len: the loop's draft length (the clamped one; smart mode: tokens after the key token)."""
    summary = np.asarray(summary).astype(np.int64)
    B = summary.shape[0]
    T_s, status = summary[:, 0], summary[:, 1]
    if (status != BP_DONE).any():
        return BeamBatchReplay(True)           # the reference asserts / the guard trips / a row nears max_len: as given
    T = int(T_s.max())
    tl = np.asarray(trace_len).astype(np.int64)[:, :T]
    t_idx = np.arange(T)[None, :]
    alive = t_idx < T_s[:, None]                                               # source b takes part in iteration t + 1 on its own
    last = tl[np.arange(B), T_s - 1][:, None]
    longest = np.where(alive, tl, last)                                        # a finished source keeps its rows
    if (longest[alive] < 1).any():
        raise ValueError("source trace shorter than its iteration count")
    batch_longest = longest.max(axis=0)                                        # after iteration t + 1
    # every iteration but the last is followed by another one, whose draft must still be draft_len long (:476)
    if T > 1 and (batch_longest[:-1] > max_len - 1 - draft_len).any():
        return BeamBatchReplay(True)
    if smart:
        tg = np.asarray(trace_grp).astype(np.int64)[:, :T]
        grp = np.where(alive, tg & 0x7f, 1)                                     # finished source: one draft per row
        sens = alive & ((tg & 0x80) != 0)
        batch_grp = grp.max(axis=0)[None, :]
        if (sens & (grp != batch_grp)).any():
            return BeamBatchReplay(True)
    width, prev = 1, 1
    for t in range(T):
        width = max(width, prev + draft_len + 1)
        prev = int(batch_longest[t])
    acc = int(summary[:, 4].sum())
    return BeamBatchReplay(False, T, width, acc, acc + int(summary[:, 5].sum()),
                           int(summary[:, 2].sum() + ((T - T_s) * n_best).sum()), int(summary[:, 3].sum()))
