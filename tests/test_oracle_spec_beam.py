"""oracle.spec_beam against the reference's beam-speculative outputs (both draft modes) and helper goldens."""
import numpy as np
import pytest
import torch

from oracle.model import OracleTransformer, config_from_state
from oracle.spec_beam import BeamSearchSpeculativeOracle, nucleus_mask, topk_per_group
from util_models import load_npz, fixture_tokens, tiny_state, PAD, BOS, EOS


@pytest.fixture(scope="module")
def model():
    st, cfg = tiny_state()
    return OracleTransformer(config_from_state(st, cfg["num_heads"]), st)


def test_nucleus_mask_matches_reference():
    g = load_npz("helpers.npz")
    x = torch.from_numpy(g["nuc_in"])
    for nucleus, nbest, num, tag in ((0.9975, 5, "-inf", "a"), (20.0, 5, 0.0, "b"), (0.9975, 10, "-inf", "c"), (0.5, 3, 0.0, "d")):
        np.testing.assert_array_equal(nucleus_mask(x.clone(), nucleus, nbest, num).numpy(), g[f"nuc_out_{tag}"])


def test_topk_per_group_matches_reference():
    g = load_npz("helpers.npz")
    s, i = topk_per_group(torch.from_numpy(g["topk_score_in"]), g["topk_lens"], 3, pad=-float("inf"))
    np.testing.assert_array_equal(s.numpy(), g["topk_score_out"])
    np.testing.assert_array_equal(i.numpy(), g["topk_idx_out"])
    s, i = topk_per_group(torch.from_numpy(g["topk_score_in"][:18]), g["topk2_lens"], 2, pad=-float("inf"))
    np.testing.assert_array_equal(s.numpy(), g["topk2_score_out"])
    np.testing.assert_array_equal(i.numpy(), g["topk2_idx_out"])


@pytest.mark.parametrize("smart", [False, True])
def test_generate_matches_reference(model, smart):
    gold = load_npz("gen_spec_beam.npz")
    src, _, c, V = fixture_tokens()
    ci = 0
    while f"smart{int(smart)}_case{ci}_rows" in gold:
        key = f"smart{int(smart)}_case{ci}"
        rows = gold[key + "_rows"].tolist()
        bsz, nbest, N, D = gold[key + "_params"].tolist()
        g = BeamSearchSpeculativeOracle(model, 150, nbest, D, N, V, smart, PAD, BOS, EOS, c, max_steps=400)
        for bi, i in enumerate(range(0, len(rows), bsz)):
            sel = src[rows[i:i + bsz]]
            width = int((sel != PAD).sum(1).max())
            out = g.generate(sel[:, :width]).numpy()
            np.testing.assert_array_equal(out, gold[f"{key}_batch{bi}"], err_msg=f"{key} batch {bi}")
        assert g.model_calls_num == int(gold[key + "_calls"]), key
        assert g.accepted_tokens_num == int(gold[key + "_accepted"]), key
        assert g.produced_non_pad_tokens == int(gold[key + "_produced"]), key
        ci += 1
    assert ci >= 5


@pytest.mark.parametrize("smart", [False, True])
def test_six_plus_six_layers_c4_shape_matches_reference(smart):
    """6+6 layers at config C4's generator settings (bs 8, n_best 10, n_drafts 2, draft_len 10, max_len 200) and two
    neighbours: the oracle against the reference's outputs in tests/golden/gen_spec_beam66.npz."""
    import json
    from util_models import GOLDEN
    st = load_npz("tiny66_weights.npz")
    cfg = json.loads((GOLDEN / "tiny66_config.json").read_text())
    m = OracleTransformer(config_from_state(st, cfg["num_heads"]), st)
    assert m.cfg["num_decoder_layers"] == 6 if isinstance(getattr(m, "cfg", None), dict) else True
    gold = load_npz("gen_spec_beam66.npz")
    src, _, c, V = fixture_tokens()
    ci = 0
    while f"smart{int(smart)}_case{ci}_rows" in gold:
        key = f"smart{int(smart)}_case{ci}"
        rows = gold[key + "_rows"].tolist()
        bsz, nbest, N, D = gold[key + "_params"].tolist()
        g = BeamSearchSpeculativeOracle(m, 200, nbest, D, N, V, smart, PAD, BOS, EOS, c, max_steps=400)
        for bi, i in enumerate(range(0, len(rows), bsz)):
            sel = src[rows[i:i + bsz]]
            width = int((sel != PAD).sum(1).max())
            np.testing.assert_array_equal(g.generate(sel[:, :width]).numpy(), gold[f"{key}_batch{bi}"], err_msg=f"{key} batch {bi}")
        assert g.model_calls_num == int(gold[key + "_calls"]), key
        assert g.accepted_tokens_num == int(gold[key + "_accepted"]), key
        assert g.produced_non_pad_tokens == int(gold[key + "_produced"]), key
        assert g.model_input_lines_num == int(gold[key + "_lines"]), key
        ci += 1
    assert ci == 3
