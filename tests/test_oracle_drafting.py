"""oracle.drafting.make_drafts against the reference's outputs (values, not only shapes)."""
import numpy as np
import pytest

from oracle.drafting import make_drafts
from util_models import load_npz, fixture_tokens, PAD, EOS

LENS = [1, 2, 3, 4, 5, 8, 10, 15, 25, 35, 50, 80, 100, 200]
AMTS = [1, 2, 3, 5, 10, 15, 25, 35, 50, 80, 100, 200]


@pytest.fixture(scope="module")
def gold():
    return load_npz("drafts.npz")


@pytest.mark.parametrize("bsz", [1, 3, 10])
def test_grid_of_reference_test_drafting(gold, bsz):
    src, _, c, _ = fixture_tokens()
    for D in LENS:
        for N in AMTS:
            got = make_drafts(src[:bsz], D, N, 1, 200, EOS, PAD, c).numpy()
            assert got.shape == (bsz, N, D)          # what tests/test_drafting.py:59-61 asserts
            np.testing.assert_array_equal(got, gold[f"full_b{bsz}_d{D}_n{N}"])


def test_generator_call_shapes(gold):
    src, _, c, _ = fixture_tokens()
    for D in (3, 5, 10, 17):
        for N in (1, 2, 3, 7, 23):
            np.testing.assert_array_equal(make_drafts(src[:, 1:], D, N, 1, 200, EOS, PAD, c).numpy(),
                                          gold[f"nobos_d{D}_n{N}"])
            np.testing.assert_array_equal(make_drafts(src[:, 1:], D, N, 5, 200, EOS, PAD, c).numpy(),
                                          gold[f"nobos_min5_d{D}_n{N}"])
    for D in (5, 10):
        np.testing.assert_array_equal(make_drafts(src, D + 1, src.shape[1] - 5, 5, 200, EOS, PAD, c).numpy(),
                                      gold[f"smartlib_d{D}"])


def test_argument_checks():
    src, _, c, _ = fixture_tokens()
    with pytest.raises(AssertionError):
        make_drafts(src, 5, 0, 1, 200, EOS, PAD, c)
    with pytest.raises(AssertionError):
        make_drafts(src, 5, 2, 1, 200, EOS, PAD, PAD)
