"""The oracle's generators against token outputs of the reference's generators on the tiny trained model."""
import numpy as np
import pytest
import torch

from oracle.model import OracleTransformer, config_from_state
from oracle.decoding import GreedyOracle, BeamSearchOracle, GreedySpeculativeOracle
from util_models import load_npz, fixture_tokens, tiny_state, PAD, BOS, EOS


@pytest.fixture(scope="module")
def model():
    st, cfg = tiny_state()
    return OracleTransformer(config_from_state(st, cfg["num_heads"]), st)


def test_greedy_matches_reference(model):
    gold = load_npz("gen_greedy.npz")
    src, _, _, _ = fixture_tokens()
    for bsz in (1, 4, 10):
        for max_len in (150, 40):
            g = GreedyOracle(model, max_len, PAD, BOS, EOS)
            for i in range(0, 10, bsz):
                out = g.generate(src[i:i + bsz]).numpy()
                ref = gold[f"b{bsz}_m{max_len}_tokens"][i:i + bsz]
                np.testing.assert_array_equal(out, ref[:, :, :out.shape[2]])
            assert g.model_calls_num == int(gold[f"b{bsz}_m{max_len}_calls"])


def test_beam_matches_reference(model):
    gold = load_npz("gen_beam.npz")
    src, _, _, _ = fixture_tokens()
    for bsz, beam in ((1, 5), (4, 5), (10, 3), (5, 10)):
        g = BeamSearchOracle(model, beam, 150, PAD, BOS, EOS)
        for bi, i in enumerate(range(0, 10, bsz)):
            out = g.generate(src[i:i + bsz]).numpy()
            np.testing.assert_array_equal(out, gold[f"b{bsz}_k{beam}_batch{bi}"])
        assert g.model_calls_num == int(gold[f"b{bsz}_k{beam}_calls"])


@pytest.mark.parametrize("bsz", [1, 4, 10])
def test_greedy_speculative_matches_reference(model, bsz):
    gold = load_npz("gen_spec_greedy.npz")
    src, _, c, _ = fixture_tokens()
    for N in (1, 3, 7, 23):
        for D in (5, 10, 17):
            g = GreedySpeculativeOracle(model, 150, D, N, PAD, BOS, EOS, c)
            out = np.concatenate([g.generate(src[i:i + bsz]).numpy() for i in range(0, 10, bsz)])
            np.testing.assert_array_equal(out, gold[f"b{bsz}_n{N}_d{D}_tokens"])
            assert g.model_calls_num == int(gold[f"b{bsz}_n{N}_d{D}_calls"])


def test_greedy_speculative_unfinished_rows_stay_pad(model):
    gold = load_npz("gen_spec_greedy.npz")
    src, _, c, _ = fixture_tokens()
    for max_len in (30, 45):
        g = GreedySpeculativeOracle(model, max_len, 10, 3, PAD, BOS, EOS, c)
        out = g.generate(src).numpy()
        np.testing.assert_array_equal(out, gold[f"short_m{max_len}_tokens"])
        assert g.model_calls_num == int(gold[f"short_m{max_len}_calls"])


def test_greedy_speculative_equals_greedy(model):
    src, _, c, _ = fixture_tokens()
    ref = GreedyOracle(model, 150, PAD, BOS, EOS).generate(src).numpy()[:, 0]
    out = GreedySpeculativeOracle(model, 150, 10, 3, PAD, BOS, EOS, c).generate(src).numpy()[:, 0]
    for a, b in zip(ref, out):
        n = int(np.argmax(a == EOS)) + 1
        np.testing.assert_array_equal(a[:n], b[:n])
