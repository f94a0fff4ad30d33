"""Host-side candidate bookkeeping of the beam generators (translation-transformer_amd/decoding.py), run on
CPU with the oracle model standing in for the HIP forward, against the reference's golden outputs."""
import numpy as np
import pytest
import torch

import translation_transformer_amd  # noqa: F401  (registers the package)
from translation_transformer_amd.decoding import _BeamSearchHost, _BeamSearchSpeculativeHost
from oracle.model import OracleTransformer, config_from_state
from util_models import load_npz, fixture_tokens, tiny_state, PAD, BOS, EOS


@pytest.fixture(scope="module")
def model():
    st, cfg = tiny_state()
    return OracleTransformer(config_from_state(st, cfg["num_heads"]), st)


def test_beam_search_host_logic(model):
    gold = load_npz("gen_beam.npz")
    src, _, _, _ = fixture_tokens()
    for bsz, beam in ((1, 5), (4, 5), (10, 3), (5, 10)):
        g = _BeamSearchHost(model, beam, 150, PAD, BOS, EOS)
        for bi, i in enumerate(range(0, 10, bsz)):
            np.testing.assert_array_equal(g.generate(src[i:i + bsz]).numpy(), gold[f"b{bsz}_k{beam}_batch{bi}"])
        assert g.model_calls_num == int(gold[f"b{bsz}_k{beam}_calls"])


@pytest.mark.parametrize("smart", [False, True])
def test_beam_speculative_host_logic(model, smart):
    gold = load_npz("gen_spec_beam.npz")
    src, _, c, V = fixture_tokens()
    ci = 0
    while f"smart{int(smart)}_case{ci}_rows" in gold:
        key = f"smart{int(smart)}_case{ci}"
        rows = gold[key + "_rows"].tolist()
        bsz, nbest, N, D = gold[key + "_params"].tolist()
        g = _BeamSearchSpeculativeHost(model, 150, nbest, D, N, V, smart, PAD, BOS, EOS, c, max_steps=400)
        for bi, i in enumerate(range(0, len(rows), bsz)):
            sel = src[rows[i:i + bsz]]
            width = int((sel != PAD).sum(1).max())
            np.testing.assert_array_equal(g.generate(sel[:, :width]).numpy(), gold[f"{key}_batch{bi}"], err_msg=key)
        assert g.model_calls_num == int(gold[key + "_calls"])
        assert g.accepted_tokens_num == int(gold[key + "_accepted"])
        assert g.produced_non_pad_tokens == int(gold[key + "_produced"])
        ci += 1
