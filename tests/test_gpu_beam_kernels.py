"""The beam-speculative bookkeeping kernels (ttx_nucleus_mask, ttx_accepted_lengths, ttx_ragged_topk) against the
oracle's restatements, which are pinned by the reference's own outputs (tests/golden/helpers.npz)."""
import numpy as np
import pytest
import torch

from oracle.spec_beam import nucleus_mask, topk_per_group
from util_models import load_npz, tiny_state

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    import translation_transformer_amd as t
    st, cfg = tiny_state()
    return t.NativeTransformer(st, cfg["num_heads"], 0, device=0)


def test_nucleus_mask_matches_reference_goldens(native):
    g = load_npz("helpers.npz")
    x = torch.from_numpy(g["nuc_in"])
    for nucleus, nbest, num, tag in ((0.9975, 5, float("-inf"), "a"), (20.0, 5, 0.0, "b"), (0.9975, 10, float("-inf"), "c"),
                                     (0.5, 3, 0.0, "d")):
        got = native.nucleus_mask(x.cuda(), nucleus, nbest, num).cpu().numpy()
        np.testing.assert_array_equal(got, g[f"nuc_out_{tag}"])


def _threshold_margin(logits: torch.Tensor, nucleus: float, nbest: int) -> torch.Tensor:
    """Per distribution: how close the "mass ranked above" of any of the first nbest ranks comes to the nucleus threshold
    (fp64).  A keep/drop decision can only differ between two correct fp32 evaluations where this is ~1e-6."""
    srt = torch.sort(logits.double(), descending=True, dim=-1).values
    above = torch.cumsum(srt.softmax(-1), dim=-1)[..., :nbest]
    return (above - nucleus).abs().min(dim=-1).values


def test_nucleus_mask_random_matches_oracle(native):
    """Random distributions: identical to the oracle everywhere, except where a rank's mass-above sits within 1e-5 of the
    threshold in exact arithmetic (the fp32 sums of the two implementations may then fall on different sides)."""
    rng = np.random.default_rng(5)
    for V in (30, 256, 1000):
        x = torch.from_numpy((rng.standard_normal((97, 11, V)) * 4).astype(np.float32))
        x[3, 2, :] = 0.0
        x[3, 2, 0] = 35.0                       # the "finished row" distribution of the beam loops
        for nucleus, nbest, fill in ((0.9975, 5, float("-inf")), (20.0, 10, 0.0), (0.9, 20, float("-inf"))):
            want = nucleus_mask(x.clone(), nucleus, nbest, fill)
            got = native.nucleus_mask(x.cuda(), nucleus, nbest, fill).cpu()
            same = (got == want) | (torch.isinf(got) & torch.isinf(want))
            bad = ~same.all(-1)
            if bool(bad.any()):
                margin = _threshold_margin(x, nucleus, nbest)[bad]
                print(f"V={V} nucleus={nucleus}: {int(bad.sum())} distribution(s) differ, threshold margin {margin.tolist()}")
                assert float(margin.max()) < 1e-5, (V, nucleus, nbest)


def test_accepted_lengths_matches_oracle(native):
    rng = np.random.default_rng(6)
    V, D, R = 64, 10, 300
    logits = torch.from_numpy((rng.standard_normal((R, D + 1, V)) * 3).astype(np.float32))
    drafts = logits[:, :-1, :].argmax(-1)                       # drafts that follow the argmax for a while ...
    flip = torch.from_numpy(rng.random((R, D)) < 0.15)
    drafts = torch.where(flip, torch.from_numpy(rng.integers(0, V, (R, D))), drafts)   # ... with random deviations
    probs = nucleus_mask(logits.clone(), 0.9975, 5, "-inf").softmax(-1)
    alive = probs[:, :-1, :].gather(2, drafts.unsqueeze(-1)).squeeze(-1) != 0
    want = alive.long().cumprod(1).sum(1)
    got = native.accepted_lengths(logits.cuda(), drafts.cuda(), 0.9975, 5).cpu()
    bad = got != want
    if bool(bad.any()):           # only rows holding a distribution whose threshold decision is within fp32 noise may differ
        margin = _threshold_margin(logits, 0.9975, 5).min(dim=-1).values[bad]
        print(f"{int(bad.sum())} draft row(s) differ, threshold margin {margin.tolist()}")
        assert float(margin.max()) < 1e-5
    assert int(want.max()) > 3 and int(want.min()) == 0


def test_ragged_topk_matches_reference_goldens_and_oracle(native):
    g = load_npz("helpers.npz")
    for lens_key, k, s_key, i_key, n in (("topk_lens", 3, "topk_score_out", "topk_idx_out", 23),
                                         ("topk2_lens", 2, "topk2_score_out", "topk2_idx_out", 18)):
        score = torch.from_numpy(g["topk_score_in"][:n]).reshape(-1)
        top, idx = native.ragged_topk(score.cuda(), torch.from_numpy(g[lens_key]).cuda(), k)
        np.testing.assert_array_equal(top.cpu().numpy(), g[s_key])
        np.testing.assert_array_equal(idx.cpu().numpy(), g[i_key])
    rng = np.random.default_rng(9)
    lens = torch.from_numpy(rng.integers(5, 400, size=37))
    score = torch.from_numpy(rng.standard_normal(int(lens.sum())).astype(np.float32))
    want_s, want_i = topk_per_group(score.clone(), lens.numpy(), 5, pad=-float("inf"))
    got_s, got_i = native.ragged_topk(score.cuda(), lens.cuda(), 5)
    assert torch.equal(got_s.cpu(), want_s) and torch.equal(got_i.cpu(), want_i)
