"""Host logic of the Lightning module's look-ahead (translation-transformer_amd/lightning_model.py:_PredictAhead) on the CPU
with a stand-in generator: windows are decoded ahead, predict_step's batches are matched by index AND content, a batch on which
the reference raises and any mismatch fall back to decoding on the spot."""
import torch

import translation_transformer_amd  # noqa: F401
from translation_transformer_amd.lightning_model import _PredictAhead


class FakeGen:
    """Counts 5 model calls per decoded batch, like a generator's model_calls_num."""

    def __init__(self, fail_on=()):
        self.calls = []
        self.fail_on = set(fail_on)
        self.model_calls_num = 0
        self.last_batch_counters = []

    def generate_many(self, batches, in_flight=4, on_error="raise"):
        assert on_error == "skip"          # a failing batch must not end the window for the others (reference: error at ITS predict_step)
        self.calls.append(len(batches))
        ok = [int(b[0, 0]) not in self.fail_on for b in batches]
        self.last_batch_counters = [{"model_calls_num": 5} if o else None for o in ok]
        self.model_calls_num += 5 * sum(ok)
        return [b.unsqueeze(1) * 10 if o else None for b, o in zip(batches, ok)]


def _loader(n):
    return [{"src_tokens": torch.full((2, 3), i, dtype=torch.int64)} for i in range(n)]


def test_windows_are_decoded_ahead_and_served_in_order():
    g = FakeGen()
    loader = _loader(7)
    ah = _PredictAhead(g, loader, window=3, in_flight=2)
    for i, b in enumerate(loader):
        out = ah.take(b["src_tokens"], i)
        assert out is not None and torch.equal(out, b["src_tokens"].unsqueeze(1) * 10)
    assert g.calls == [3, 3, 1] and ah.served == 7 and ah.fallbacks == 0 and ah.windows == 3
    assert ah.take(loader[0]["src_tokens"], 7) is None         # past the end of the dataloader: nothing prepared


def test_batch_on_which_the_reference_raises_is_left_to_generate():
    g = FakeGen(fail_on={2})
    loader = _loader(5)
    ah = _PredictAhead(g, loader, window=5, in_flight=2)
    got = [ah.take(b["src_tokens"], i) for i, b in enumerate(loader)]
    assert got[2] is None and all(got[i] is not None for i in (0, 1, 3, 4))
    assert ah.enabled and ah.served == 4 and ah.fallbacks == 1
    assert g.model_calls_num == 20                             # the failing batch was never counted: generate() raises it


def test_other_batches_than_the_dataloaders_switch_the_look_ahead_off():
    g = FakeGen()
    loader = _loader(4)
    ah = _PredictAhead(g, loader, window=2, in_flight=2)
    assert ah.take(loader[0]["src_tokens"], 0) is not None
    assert ah.take(loader[3]["src_tokens"], 1) is None         # index 1 was prepared from other tokens
    assert not ah.enabled
    assert ah.take(loader[2]["src_tokens"], 2) is None and g.calls == [2]
    # batch 1 was decoded ahead but is decoded again by generate(): its share of the counters is taken back
    assert g.model_calls_num == 5
    # a different shape is a mismatch as well
    ah2 = _PredictAhead(FakeGen(), loader, window=2, in_flight=1)
    assert ah2.take(torch.zeros((2, 4), dtype=torch.int64), 0) is None and not ah2.enabled


def test_one_shot_iterators_are_not_looked_ahead_on():
    """The look-ahead walks the predict dataloader a second time: a loader that is its own iterator would be consumed."""
    from types import SimpleNamespace
    from translation_transformer_amd.lightning_model import VanillaEncoderDecoderTransformerLightning as M
    tk = SimpleNamespace(pad_token_idx=0, bos_token_idx=1, eos_token_idx=2, n_tokens=16, encoder_dict={"c": 4})
    mod = M(src_tokenizer=tk, tgt_tokenizer=tk, embedding_dim=64, feedforward_dim=64, num_encoder_layers=1, num_decoder_layers=1,
            num_heads=2, generation="greedy_speculative", max_len=20, n_drafts=1, draft_len=3, report_prediction_time=False)
    mod.build_native = lambda *a, **k: None
    mod.generator = FakeGen()
    mod.trainer = SimpleNamespace(datamodule=None, predict_dataloaders=iter(_loader(3)))
    mod.on_predict_start()
    assert mod._ahead is None
    mod.trainer = SimpleNamespace(datamodule=None, predict_dataloaders=_loader(3))
    mod.on_predict_start()
    assert mod._ahead is not None
