"""Native greedy-speculative generator (ttx_greedy_speculative_generate) against the reference's token
outputs (golden) and against the oracle on the same weights."""
import numpy as np
import pytest
import torch

from util_models import load_npz, tiny_state, fixture_tokens, PAD, BOS, EOS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tta():
    import translation_transformer_amd as t
    assert t.lib().ttx_device_count() >= 1
    return t


@pytest.fixture(scope="module")
def tiny(tta):
    st, cfg = tiny_state()
    return tta.NativeTransformer(st, cfg["num_heads"], 0, device=0)


@pytest.mark.parametrize("bsz", [1, 4, 10])
def test_matches_reference_tokens(tta, tiny, bsz):
    gold = load_npz("gen_spec_greedy.npz")
    src, _, c, _ = fixture_tokens()
    for N in (1, 3, 7, 23):
        for D in (5, 10, 17):
            g = tta.TranslationInferenceGreedySpeculative(tiny, 150, D, N, PAD, BOS, EOS, c)
            out = np.concatenate([g.generate(src[i:i + bsz].cuda()).cpu().numpy() for i in range(0, 10, bsz)])
            np.testing.assert_array_equal(out, gold[f"b{bsz}_n{N}_d{D}_tokens"], err_msg=f"b{bsz} n{N} d{D}")
            assert g.model_calls_num == int(gold[f"b{bsz}_n{N}_d{D}_calls"])


def test_unfinished_rows_stay_pad(tta, tiny):
    gold = load_npz("gen_spec_greedy.npz")
    src, _, c, _ = fixture_tokens()
    for max_len in (30, 45):
        g = tta.TranslationInferenceGreedySpeculative(tiny, max_len, 10, 3, PAD, BOS, EOS, c)
        out = g.generate(src.cuda()).cpu().numpy()
        np.testing.assert_array_equal(out, gold[f"short_m{max_len}_tokens"])
        assert g.model_calls_num == int(gold[f"short_m{max_len}_calls"])


def test_many_batches_in_flight_equal_one_at_a_time(tta, tiny):
    src, _, c, _ = fixture_tokens()
    g1 = tta.TranslationInferenceGreedySpeculative(tiny, 150, 10, 3, PAD, BOS, EOS, c)
    batches = []
    for lo, hi in ((0, 3), (3, 4), (4, 8), (8, 10), (0, 10), (5, 9)):
        sel = src[lo:hi]
        batches.append(sel[:, :int((sel != PAD).sum(1).max())].cuda())
    ref = [g1.generate(b) for b in batches]
    for in_flight in (1, 2, 4):
        g2 = tta.TranslationInferenceGreedySpeculative(tiny, 150, 10, 3, PAD, BOS, EOS, c)
        out = g2.generate_many(batches, in_flight=in_flight)
        assert len(out) == len(ref)
        for a, b in zip(out, ref):
            assert torch.equal(a, b)
        assert g2.model_calls_num == g1.model_calls_num
