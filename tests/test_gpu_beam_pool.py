"""The beam-speculative batch pool (ttx_beam_speculative_generate_pool behind generate_many(pool=True)): many given batches
decoded in slot pools (whole batches admitted as slots free up, one verify step per iteration for all of them, the reference's
batch-wide loop scalars kept per batch on the device), result width / model calls / counters replayed from per-source traces.
Everything the pooled call returns — every hypothesis tensor including its width, and every counter — must equal what
per-batch ``generate`` calls return (which the golden / oracle tests of test_gpu_beam_native.py pin to the reference)."""
import numpy as np
import pytest
import torch

from util_models import tiny_state, fixture_tokens, PAD, BOS, EOS

pytestmark = pytest.mark.gpu

COUNTERS = ("model_calls_num", "accepted_tokens_num", "produced_non_pad_tokens", "model_input_lines_num", "b_sz", "n_drafts")


@pytest.fixture(scope="module")
def tta():
    import translation_transformer_amd as t
    assert t.lib().ttx_device_count() >= 1
    return t


def _batches(src, groups):
    out = []
    for rows in groups:
        sel = src[rows]
        out.append(sel[:, :int((sel != PAD).sum(1).max())].cuda())
    return out


def _same(tta, native, batches, params, smart, **kw):
    max_len, nbest, D, N = params
    _, _, c, V = fixture_tokens()
    one = tta.TranslationInferenceBeamSearchSpeculative(native, max_len, nbest, D, N, V, smart, PAD, BOS, EOS, c, max_steps=300)
    ref = [one.generate(b) for b in batches]
    many = tta.TranslationInferenceBeamSearchSpeculative(native, max_len, nbest, D, N, V, smart, PAD, BOS, EOS, c, max_steps=300)
    out = many.generate_many(batches, in_flight=3, pool=True, **kw)
    for i, (a, b) in enumerate(zip(out, ref)):
        assert a.shape == b.shape, (i, a.shape, b.shape, params, smart)
        assert torch.equal(a, b), (i, params, smart)
    for name in COUNTERS:
        assert getattr(many, name) == getattr(one, name), (name, params, smart)
    return many


@pytest.mark.parametrize("smart", [False, True])
def test_pooled_sources_equal_per_batch_calls_tiny_model(tta, smart):
    """Tiny 2+2 model (2 heads: the step runs on k_attn2), ragged batches of 1-5 sources, several settings; small pools so that
    slots are re-used by later sources while earlier ones are still running."""
    st, cfg = tiny_state()
    native = tta.NativeTransformer(st, cfg["num_heads"], 0, device=0)
    src, _, _, _ = fixture_tokens()
    groups = [[0, 2, 3, 4], [5, 6], [8, 9, 0], [2], [3, 4, 5, 6, 8], [9, 2], [6, 0, 4], [8]]
    batches = _batches(src, groups)
    for params in ((150, 5, 10, 3), (150, 3, 5, 7), (200, 10, 10, 2), (150, 1, 10, 2)):
        for cap in (3, 64):
            m = _same(tta, native, batches, params, smart, capacity=cap)
            assert m.stats_total.get("pool_calls", 0) == 1


def test_draft_length_shrinking_near_max_len_is_handled_in_the_pool(tta):
    """max_len so small that hypotheses come within draft_len + 1 of it: the draft length shrinks per batch (:476) and the loop
    may end for lack of room (:464) with unfinished hypotheses — batch-wide scalars the pool keeps per batch on the device."""
    st, cfg = tiny_state()
    native = tta.NativeTransformer(st, cfg["num_heads"], 0, device=0)
    src, _, _, _ = fixture_tokens()
    batches = _batches(src, [[0, 2, 4], [3, 5], [2, 6, 8, 9], [9]])
    for smart in (False, True):
        for params in ((33, 3, 10, 3), (9, 2, 10, 2), (20, 5, 10, 2), (12, 3, 17, 3)):
            for cap in (4, 64):
                m = _same(tta, native, batches, params, smart, capacity=cap)
                assert m.stats_total.get("pool_calls", 0) == 1


def test_pool_error_batches(tta):
    """A source on which the loop does not terminate (fixture row 1 at n_best 5: the max_steps guard) sits in one of several
    batches: on_error='skip' yields None for that batch only, the others equal per-batch calls; on_error='raise' raises."""
    st, cfg = tiny_state()
    native = tta.NativeTransformer(st, cfg["num_heads"], 0, device=0)
    src, _, c, V = fixture_tokens()
    batches = _batches(src, [[0, 2], [1, 3], [4, 5, 6]])
    mk = lambda: tta.TranslationInferenceBeamSearchSpeculative(native, 150, 5, 10, 7, V, False, PAD, BOS, EOS, c, max_steps=60)
    g = mk()
    out = g.generate_many(batches, pool=True, on_error="skip")
    assert out[1] is None and g.last_failed_batches == [1]
    one = mk()
    assert torch.equal(out[0], one.generate(batches[0])) and torch.equal(out[2], one.generate(batches[2]))
    assert g.model_calls_num == one.model_calls_num and g.accepted_tokens_num == one.accepted_tokens_num
    with pytest.raises(RuntimeError, match="max_steps"):
        mk().generate_many(batches, pool=True)
    with pytest.raises(RuntimeError, match="max_steps"):
        mk().generate(batches[1])


@pytest.mark.parametrize("layers,params", [(4, (200, 5, 10, 7)), (6, (200, 10, 10, 2))])
def test_pooled_sources_equal_per_batch_calls_full_size(tta, trained_full_state, layers, params):
    """Configs C3 (4+4, n_best 5, N 7, bs 4) and C4 (6+6, n_best 10, N 2, bs 8) at their real layer sizes (8 heads: the step
    runs on k_attn3), both draft modes: pooled == per batch, bit for bit — the pool's large-row GEMM variant and the per-batch
    path's small-row variant evaluate the same ordered slice sums (csrc/ttx_gemm.hip)."""
    st = trained_full_state(layers)
    native = tta.NativeTransformer(st, 8, 0, device=0)
    src, _, _, _ = fixture_tokens()
    bs = 4 if layers == 4 else 8
    rows = [0, 2, 3, 4, 5, 6, 8, 9]
    groups = [[rows[(i + j) % 8] for j in range(bs)] for i in (0, 3, 5)] + [[2, 9]]
    batches = _batches(src, groups)
    for smart in (False, True):
        m = _same(tta, native, batches, params, smart)
        print(f"{layers}+{layers} smart={smart}: device iterations", m.stats_total.get("device_model_calls"), "for", m.model_calls_num,
              "replayed calls")
