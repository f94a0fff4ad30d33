"""CPU test of the host half of the beam-speculative batch pool (translation-transformer_amd/scheduling.py
replay_beam_batch).  The pool keeps the batch-wide loop scalars on the device and returns per-source traces; the replay
derives the result width, the model calls and the counters of every given batch from them.  Here the traces are taken from the
oracle's own run of the batch (oracle.spec_beam with ``trace``), cut per source exactly where the pool retires a source (every
one of its rows holds EOS), and the replay must reproduce what the oracle reports for the batch — including the input lines a
finished source keeps contributing and the width rule when the draft length shrinks near max_len."""
import numpy as np
import pytest

from util_models import tiny_state, fixture_tokens, PAD, BOS, EOS


def _pool_arrays(trace, B, nbest, max_len):
    """What ttx_beam_speculative_generate_pool would return for this batch: a source retires at the first iteration after
    which all its rows hold EOS (status 1), otherwise with its batch (status 3)."""
    T = len(trace)
    T_cap = max_len + 8
    tl = np.full((B, T_cap), -1, np.int16)
    sm = np.zeros((B, 8), np.int32)
    for b in range(B):
        done = [t for t, r in enumerate(trace) if int(r["n_eos"][b]) == nbest]
        T_s = done[0] + 1 if done else T
        for t in range(T_s):
            tl[b, t] = trace[t]["longest"][b]
        part = trace[:T_s]
        sm[b] = [T_s, 1 if done else 3, sum(int(r["lines"][b]) for r in part), sum(int(r["running"][b]) for r in part),
                 sum(int(r["acc_sum"][b]) for r in part), sum(int(r["acc_cnt"][b]) for r in part), int(part[-1]["longest"][b]),
                 sum(int(r["run_cands"][b]) for r in part)]
    return tl, sm


@pytest.mark.parametrize("smart", [False, True])
def test_replay_reproduces_the_oracles_batch_bookkeeping(smart):
    from oracle.model import OracleTransformer, config_from_state
    from oracle.spec_beam import BeamSearchSpeculativeOracle
    from translation_transformer_amd.scheduling import replay_beam_batch
    st, cfg = tiny_state()
    oracle = OracleTransformer(config_from_state(st, cfg["num_heads"]), st)
    src, _, c, V = fixture_tokens()
    rng = np.random.default_rng(77 + int(smart))
    checked = shrunk = 0
    for trial in range(8):
        rows = rng.choice([0, 2, 3, 4, 5, 6, 8, 9], size=int(rng.integers(1, 5)), replace=False).tolist()
        nbest = int(rng.choice([2, 3, 5]))
        N = int(rng.choice([2, 3, 7]))
        D = int(rng.choice([5, 10]))
        max_len = int(rng.choice([150, 200, 60, 33]))
        sel = src[rows]
        sel = sel[:, :int((sel != PAD).sum(1).max())]
        ref = BeamSearchSpeculativeOracle(oracle, max_len, nbest, D, N, V, smart, PAD, BOS, EOS, c, max_steps=150)
        ref.trace = []
        try:
            exp = ref.generate(sel).numpy()
        except (RuntimeError, AssertionError):
            continue
        d0 = ref.draft_len if not smart else min(max(5, ref.draft_len + 1), 200) - 1
        tl, sm = _pool_arrays(ref.trace, len(rows), nbest, max_len)
        rep = replay_beam_batch(tl, sm, max_len, d0, nbest)
        label = (trial, rows, nbest, N, D, max_len, smart)
        assert rep.error is None, label
        assert rep.out_width == exp.shape[2], label
        assert rep.model_calls == ref.model_calls_num, label
        assert rep.accepted_tokens == ref.accepted_tokens_num and rep.produced_non_pad_tokens == ref.produced_non_pad_tokens, label
        if smart:
            assert rep.input_lines == ref.model_input_lines_num, label
        checked += 1
        shrunk += int(max_len <= 60)
    print(f"smart={smart}: {checked} batches replayed, {shrunk} of them with the draft length shrinking near max_len")
    assert checked >= 5 and shrunk >= 1


def test_replay_arithmetic_and_errors():
    from translation_transformer_amd.scheduling import replay_beam_batch
    T_cap = 40
    tl = np.full((2, T_cap), -1, np.int16)
    tl[0, :3] = [6, 12, 15]
    tl[1, :5] = [5, 9, 14, 20, 22]
    sm = np.array([[3, 1, 30, 20, 9, 12, 15, 9], [5, 1, 50, 40, 12, 20, 22, 15]], np.int32)
    rep = replay_beam_batch(tl, sm, 30, 5, 2)
    assert rep.error is None and rep.model_calls == 5
    # widths: 1 -> 7, then max(7, 6 + 6) = 12, max(12, 12 + 6) = 18, max(18, 15 + 6) = 21, max(21, 20 + 6) = 26
    assert rep.out_width == 26
    assert rep.accepted_tokens == 21 and rep.produced_non_pad_tokens == 53
    # a finished source keeps contributing n_best input lines per remaining iteration of its batch
    assert rep.input_lines == 30 + 50 + (5 - 3) * 2 and rep.running_rows == 60
    # near max_len the draft shrinks: after a longest row of 26 only max_len - 26 - 1 = 3 draft tokens fit
    tl2 = tl.copy()
    tl2[1, 3:6] = [26, 28, 29]
    sm2 = sm.copy()
    sm2[1] = [6, 3, 50, 40, 12, 20, 29, 15]
    rep2 = replay_beam_batch(tl2, sm2, 30, 5, 2)
    # 1 -> 7 -> 12 -> 18 -> 21 -> max(21, 26 + 3 + 1) = 30 -> max(30, 28 + 1 + 1) = 30
    assert rep2.error is None and rep2.model_calls == 6 and rep2.out_width == 30
    for bad, name in ((2, "reference"), (4, "max_steps")):
        sm3 = sm.copy()
        sm3[1, 1] = bad
        assert replay_beam_batch(tl, sm3, 30, 5, 2).error == name
