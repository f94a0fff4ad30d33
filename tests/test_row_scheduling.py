"""Row scheduling (translation-transformer_amd/scheduling.py): replaying the reference's per-batch width rule from
per-row traces gives exactly what the greedy-speculative loop (oracle restatement of
src/decoding/speculative_decoding.py:39-174, pinned by gen_spec_greedy.npz) does on the batch as a whole.

The traces here come from the oracle run on one row at a time (a row alone in its batch IS the per-row rule);
on the GPU they come from ttx_greedy_speculative_generate_rows (tests/test_gpu_spec_greedy.py)."""
import numpy as np
import pytest
import torch

from oracle.model import OracleTransformer, config_from_state
from oracle.decoding import GreedySpeculativeOracle
from util_models import fixture_tokens, tiny_state, PAD, BOS, EOS

import translation_transformer_amd  # noqa: F401  (registers the package under its importable name)
from translation_transformer_amd.scheduling import replay_batch, plan_row_groups


@pytest.fixture(scope="module")
def model():
    st, cfg = tiny_state()
    return OracleTransformer(config_from_state(st, cfg["num_heads"]), st)


def row_trace(model, row, max_len, D, N, c_token):
    """(tokens [max_len], traj [max_len+1], fin_step) of one row decoded alone; a finish beyond max_len (which makes
    the reference raise for a batch of one) is recorded as a finish, as the library's per-row rule does."""
    g = GreedySpeculativeOracle(model, max_len, D, N, PAD, BOS, EOS, c_token)
    n = int((row != PAD).sum())
    try:
        out = g.generate(row[None, :n]).numpy()[0, 0]
        raised = False
    except RuntimeError as e:
        assert "wider than max_len" in str(e)
        out, raised = np.full(max_len, PAD, dtype=np.int64), True
    traj = np.full(max_len + 1, -1, dtype=np.int64)
    traj[0] = 0
    for t, (_, front) in enumerate(g.front_log):
        traj[t + 1] = front[0] if len(front) else -1
    finished = raised or bool((out == EOS).any())
    return out, traj, (len(g.front_log) if finished else 0)


@pytest.mark.parametrize("max_len,D,N", [(150, 4, 2), (40, 4, 2), (28, 6, 3), (24, 10, 1), (33, 3, 2), (16, 4, 2), (12, 6, 1), (14, 3, 3), (20, 8, 2), (45, 4, 2), (27, 4, 2), (57, 4, 2)])
def test_replay_equals_batch_loop(model, max_len, D, N):
    src, _, c_token, _ = fixture_tokens()
    rows = [row_trace(model, src[i], max_len, D, N, c_token) for i in range(src.shape[0])]
    n_err = n_ok = 0
    for batch in ([0, 1, 2, 3], [4, 5, 6, 7, 8, 9], [9, 0, 5], [2], [7, 3, 1, 8, 6], list(range(10))):
        g = GreedySpeculativeOracle(model, max_len, D, N, PAD, BOS, EOS, c_token)
        traj = np.stack([rows[i][1] for i in batch])
        fin = np.array([rows[i][2] for i in batch])
        rep = replay_batch(traj, fin, max_len, D, N)
        try:
            want = g.generate(src[batch]).numpy()[:, 0]
        except RuntimeError as e:
            assert "wider than max_len" in str(e)
            assert rep.error
            n_err += 1
            continue
        n_ok += 1
        assert not rep.error
        assert rep.model_calls == g.model_calls_num
        assert rep.accepted_tokens == g.accepted_total
        got = np.stack([rows[i][0] if rep.finished[k] else np.full(max_len, PAD) for k, i in enumerate(batch)])
        np.testing.assert_array_equal(got, want)
    print(f"max_len={max_len} D={D} N={N}: {n_ok} batches replayed, {n_err} raise like the reference")
    assert n_ok + n_err == 6


def test_replay_synthetic_width_rule():
    # two rows, D = 2, max_len = 12: row 0 advances 3 per step and finishes at step 3; row 1 crawls.
    traj = np.full((2, 13), -1)
    traj[0, :4] = [0, 3, 6, 9]
    traj[1, :10] = np.arange(10)        # alone, row 1 runs until the step that starts at front 8 (8 + 2 + 2 >= 12)
    fin = np.array([3, 0])
    rep = replay_batch(traj, fin, 12, 2, 1)
    # widths 4, 7, 10: the third iteration finishes row 0 at width 10 <= 12; row 1 then runs on alone up to width 12
    assert not rep.error and rep.model_calls == 9 and list(rep.finished) == [True, False]
    assert rep.rows_iterations == 3 * 2 + 6 and rep.accepted_tokens == 6
    # a finish beyond max_len raises: row finishing at its 4th step where width = 9 + 4 = 13 > 12
    traj2 = np.full((1, 13), -1)
    traj2[0, :5] = [0, 3, 6, 9, 11]
    rep2 = replay_batch(traj2, np.array([4]), 12, 2, 1)
    assert rep2.error
    # the loop ends at width >= max_len and leaves the row unfinished
    rep3 = replay_batch(traj2, np.array([0]), 12, 2, 1)
    assert not rep3.error and rep3.model_calls == 4 and not rep3.finished[0]


def test_plan_row_groups():
    order, groups = plan_row_groups([5, 9, 2, 9, 7], 2)
    assert list(order) == [1, 3, 4, 0, 2]
    assert [(g.start, g.stop) for g in groups] == [(0, 2), (2, 4), (4, 5)]


def _replay_loop(traj, fin_step, max_len, D, N=1):
    """The reference's while-loop written out iteration by iteration (what replay_batch vectorises)."""
    B = traj.shape[0]
    running = np.ones(B, dtype=bool)
    finished = np.zeros(B, dtype=bool)
    width, t, calls, acc, rows_it = 1, 0, 0, 0, 0
    while width < max_len and running.any():
        t += 1
        before, after = traj[running, t - 1], traj[running, t]
        assert (before >= 0).all() and (after >= 0).all()
        width = int(before.max()) + D + 2
        calls += 1
        rows_it += int(running.sum())
        acc += int((after - before - 1).sum())
        done = running & (fin_step == t)
        if done.any():
            if width > max_len:
                return calls, True, np.zeros(B, dtype=bool), acc, rows_it
            finished |= done
            running &= ~done
    return calls, False, finished, acc, rows_it


def test_replay_vectorised_equals_loop_on_random_traces():
    """Random per-row traces generated under the per-row rule (a row goes on while front + D + 2 < max_len), random
    finish steps (including finishes beyond max_len, which make the batch raise)."""
    rng = np.random.default_rng(2024)
    n_err = 0
    for trial in range(400):
        max_len = int(rng.integers(8, 60))
        D = int(rng.integers(1, min(12, max_len) + 1))
        B = int(rng.integers(1, 12))
        traj = np.full((B, max_len + 1), -1, dtype=np.int64)
        fin = np.zeros(B, dtype=np.int64)
        for r in range(B):
            f, t = 0, 0
            traj[r, 0] = 0
            target = int(rng.integers(1, max_len + D))            # position of the row's EOS (may be out of reach)
            while True:
                step = int(rng.integers(1, D + 2))                  # accepted + 1 bonus token
                before = f
                f = min(f + step, max_len + D)
                t += 1
                traj[r, t] = f
                if f >= target:                                     # EOS produced in this step
                    fin[r] = t
                    break
                if before + D + 2 >= max_len or t >= max_len:       # the row alone would stop here
                    break
        rep = replay_batch(traj, fin, max_len, D, 1)
        calls, err, finished, acc, rows_it = _replay_loop(traj, fin, max_len, D, 1)
        assert rep.error == err, trial
        assert rep.model_calls == calls, trial
        assert rep.rows_iterations == rows_it and rep.accepted_tokens == acc, trial
        if not err:
            np.testing.assert_array_equal(rep.finished, finished)
        n_err += err
    assert 0 < n_err < 400
