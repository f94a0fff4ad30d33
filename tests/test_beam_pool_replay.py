"""CPU test of the beam-speculative source pool's bookkeeping (translation-transformer_amd/scheduling.py
replay_beam_batch): every source of a batch is decoded ALONE by the oracle (padded to the batch's width, as the HIP pool
does), the per-source traces the pool would record are assembled from those runs, and the replay must give exactly what the
oracle does with the batch as given — hypotheses, result width, model calls and counters — or say that the batch has to be
decoded as given.  This pins the claim the pool rests on: the sources of a batch interact only through batch-wide scalars."""
import numpy as np
import pytest
import torch

from util_models import tiny_state, fixture_tokens, PAD, BOS, EOS


def _alone_traces(oracle, sel, params, smart):
    """Decode every source alone and build the arrays ttx_beam_speculative_generate_pool returns."""
    from oracle.spec_beam import BeamSearchSpeculativeOracle
    nbest, D, N, max_len = params
    _, _, c, V = fixture_tokens()
    B = sel.shape[0]
    T_cap = max_len + 8
    tl = np.full((B, T_cap), -1, np.int16)
    tg = np.zeros((B, T_cap), np.uint8)
    sm = np.zeros((B, 8), np.int32)
    rows = []
    for b in range(B):
        g = BeamSearchSpeculativeOracle(oracle, max_len, nbest, D, N, V, smart, PAD, BOS, EOS, c, max_steps=120)
        g.trace = []
        try:
            out = g.generate(sel[b:b + 1])            # padded to the batch's width: smart mode's library depends on it
            status = 1
        except RuntimeError:
            out, status = None, 4
        except AssertionError:
            out, status = None, 2
        d0 = g.draft_len if not smart else min(max(5, g.draft_len + 1), 200) - 1
        for t, rec in enumerate(g.trace):
            tl[b, t] = rec["longest"][0]
            tg[b, t] = int(rec["grp"][0]) | (0x80 if rec["sens"][0] else 0)
        if status == 1 and g.trace and int(g.trace[-1]["n_eos"][0]) != nbest:
            status = 3                                # the loop ended because no room was left: not a finished source
        # a running source whose longest row came within draft_len + 1 of max_len is retired by the pool
        for t, rec in enumerate(g.trace[:-1]):
            if rec["longest"][0] > max_len - 1 - d0:
                status = 3
        sm[b] = [len(g.trace), status, sum(int(r["lines"][0]) for r in g.trace), sum(int(r["running"][0]) for r in g.trace),
                 sum(int(r["acc_sum"][0]) for r in g.trace), sum(int(r["acc_cnt"][0]) for r in g.trace),
                 int(g.trace[-1]["longest"][0]) if g.trace else 0, sum(int(r["run_cands"][0]) for r in g.trace)]
        rows.append(out)
    return tl, tg, sm, rows, d0


@pytest.mark.parametrize("smart", [False, True])
def test_replay_of_sources_decoded_alone_equals_the_batch(smart):
    from oracle.model import OracleTransformer, config_from_state
    from oracle.spec_beam import BeamSearchSpeculativeOracle
    from translation_transformer_amd.scheduling import replay_beam_batch
    st, cfg = tiny_state()
    oracle = OracleTransformer(config_from_state(st, cfg["num_heads"]), st)
    src, _, c, V = fixture_tokens()
    rng = np.random.default_rng(77 + int(smart))
    replayed = as_given = 0
    for trial in range(8):
        rows = rng.choice([0, 2, 3, 4, 5, 6, 8, 9], size=int(rng.integers(2, 5)), replace=False).tolist()
        nbest = int(rng.choice([2, 3, 5]))
        N = int(rng.choice([2, 2, 3, 7] if smart else [2, 3, 7]))      # smart mode with many drafts often couples the sources
        D = int(rng.choice([5, 10]))
        max_len = int(rng.choice([150, 150, 200, 60]))
        sel = src[rows]
        sel = sel[:, :int((sel != PAD).sum(1).max())]
        ref = BeamSearchSpeculativeOracle(oracle, max_len, nbest, D, N, V, smart, PAD, BOS, EOS, c, max_steps=120)
        try:
            exp = ref.generate(sel).numpy()
        except (RuntimeError, AssertionError):
            exp = None
        tl, tg, sm, outs, d0 = _alone_traces(oracle, sel, (nbest, D, N, max_len), smart)
        rep = replay_beam_batch(tl, tg, sm, max_len, d0, nbest, smart)
        label = (trial, rows, nbest, N, D, max_len, smart)
        if rep.as_given:
            as_given += 1
            continue
        assert exp is not None, label
        assert rep.out_width == exp.shape[2], label
        assert rep.model_calls == ref.model_calls_num, label
        assert rep.accepted_tokens == ref.accepted_tokens_num and rep.produced_non_pad_tokens == ref.produced_non_pad_tokens, label
        if smart:
            assert rep.input_lines == ref.model_input_lines_num, label
        for b, o in enumerate(outs):
            got = torch.nn.functional.pad(o[0], (0, max(0, exp.shape[2] - o.shape[2])), value=PAD).numpy()[:, :exp.shape[2]]
            np.testing.assert_array_equal(got, exp[b], err_msg=str(label + (b,)))
            assert (o[0].numpy()[:, exp.shape[2]:] == PAD).all()
        replayed += 1
    print(f"smart={smart}: {replayed} batches replayed exactly, {as_given} sent back to be decoded as given")
    assert replayed >= 3


def test_replay_sends_coupled_batches_back():
    from translation_transformer_amd.scheduling import replay_beam_batch
    T_cap = 40
    tl = np.full((2, T_cap), -1, np.int16)
    tg = np.zeros((2, T_cap), np.uint8)
    tl[0, :3] = [6, 12, 15]
    tl[1, :5] = [5, 9, 14, 20, 22]
    sm = np.array([[3, 1, 30, 20, 9, 12, 15, 9], [5, 1, 50, 40, 12, 20, 22, 15]], np.int32)
    rep = replay_beam_batch(tl, tg, sm, 30, 5, 2, False)
    assert not rep.as_given and rep.model_calls == 5
    # widths: 1 -> 7, then max(7, 6 + 6) = 12, max(12, 12 + 6) = 18, max(18, 15 + 6) = 21, max(21, 20 + 6) = 26
    assert rep.out_width == 26
    assert rep.accepted_tokens == 21 and rep.produced_non_pad_tokens == 53
    # a finished source longer than max_len - 1 - draft_len while the other still runs: the next draft would be cut
    tl2 = tl.copy()
    tl2[0, 2] = 25
    sm2 = sm.copy()
    sm2[0, 6] = 25
    assert replay_beam_batch(tl2, tg, sm2, 30, 5, 2, False).as_given
    # any source that did not finish (error, guard, near max_len) sends the batch back
    for bad in (2, 3, 4, 5):
        sm3 = sm.copy()
        sm3[1, 1] = bad
        assert replay_beam_batch(tl, tg, sm3, 30, 5, 2, False).as_given
    # smart mode: a table-width-sensitive choice while another source has a longer group
    tg4 = tg.copy()
    tg4[0, :3] = [1, 2, 2]
    tg4[1, :5] = [1, 3 | 0x80, 3, 2, 2]
    assert not replay_beam_batch(tl, tg4, sm, 30, 5, 2, True).as_given        # the sensitive source has the longest group itself
    tg4[0, 1] = 4
    assert replay_beam_batch(tl, tg4, sm, 30, 5, 2, True).as_given
    # smart-mode input lines: a finished source keeps contributing n_best lines per remaining iteration
    assert replay_beam_batch(tl, tg, sm, 30, 5, 2, True).input_lines == 30 + 50 + (5 - 3) * 2
