"""Lightning surface of the reference kept for the inference path:

    VanillaEncoderDecoderTransformerLightning   <- src/model/lightning_model.py:22-243

Same ``init_args`` (lightning_model.py:24-51), same ``predict_step`` / ``on_predict_start`` /
``on_predict_end`` hooks and JSON report keys, same attributes the PredictionWriter callback reads
(``tgt_tokenizer``; src/callbacks.py:49-64), same checkpoint key layout (``model.`` + the names of
SURVEY §8(b) B6) — so ``main.py predict -c cfg.yaml --ckpt_path ...`` runs unchanged once the YAML's
``model.class_path`` points here.  Training hooks are out of scope and raise.

``self.model`` is a parameter container with the reference's module names (so Lightning's checkpoint loading
fills it); its torch ``forward`` is never called — all arithmetic runs in libttx_hip.so through the
NativeTransformer built from those parameters when prediction starts.  The module imports without
pytorch_lightning (absent in the build container); then a minimal stand-in base class is used and
``run_predict`` below drives the hooks.
"""
from __future__ import annotations

import datetime
import json
import os
from pathlib import Path
from timeit import default_timer as timer
from types import SimpleNamespace
from typing import Any

import torch
from torch import nn

try:  # pragma: no cover - depends on the environment
    from pytorch_lightning import LightningModule
    HAVE_LIGHTNING = True
except Exception:  # pragma: no cover
    HAVE_LIGHTNING = False

    class LightningModule(nn.Module):  # minimal stand-in: hparams + the hooks this file uses
        def __init__(self):
            super().__init__()
            self.hparams = SimpleNamespace()
            self.trainer = None

        def save_hyperparameters(self, ignore=()):
            import inspect
            frame = inspect.currentframe().f_back
            args = {k: v for k, v in frame.f_locals.items() if k not in ("self", "__class__") and k not in ignore}
            self.hparams = SimpleNamespace(**args)

from .model import NativeTransformer
from . import decoding as D


class _Emb(nn.Module):
    def __init__(self, vocab, d, pad):
        super().__init__()
        self.embedding = nn.Embedding(vocab, d, padding_idx=pad)


class WeightContainer(nn.Module):
    """Parameters only, named exactly like the reference's VanillaTransformer (src/model/modules.py:40-84)."""

    def __init__(self, src_vocab, tgt_vocab, n_enc, n_dec, d, heads, ff, share, src_pad, tgt_pad):
        super().__init__()
        self.src_pad_token_i, self.tgt_pad_token_i = src_pad, tgt_pad
        self.emb_dim, self.num_heads = d, heads
        self.src_token_featurizer = _Emb(src_vocab, d, src_pad)
        self.tgt_token_featurizer = self.src_token_featurizer if share else _Emb(tgt_vocab, d, tgt_pad)
        enc = nn.TransformerEncoder(nn.TransformerEncoderLayer(d, heads, ff, 0.0, "relu", 1e-5, True, False), n_enc,
                                    nn.LayerNorm(d, eps=1e-5), enable_nested_tensor=False)
        dec = nn.TransformerDecoder(nn.TransformerDecoderLayer(d, heads, ff, 0.0, "relu", 1e-5, True, False), n_dec,
                                    nn.LayerNorm(d, eps=1e-5))
        self.transformer = nn.Transformer(d_model=d, nhead=heads, batch_first=True, custom_encoder=enc, custom_decoder=dec)
        self.next_token_classifier = nn.Linear(d, tgt_vocab)

    def forward(self, *a, **k):
        raise RuntimeError("the torch modules here only hold parameters; the forward pass runs in libttx_hip.so")


class _PredictAhead:
    """Decodes windows of the predict dataloader ahead of ``predict_step`` so that the rows of many batches share the
    GPU (slot pools for greedy-speculative: generate_many(reorder=True); batches in flight for beam-speculative), while
    ``Trainer.predict`` / ``main.py`` keep calling ``predict_step(batch, batch_idx)`` one batch at a time.

    The reference's loop decodes batch i inside predict_step(i) (src/model/lightning_model.py:209-212).  Here the module
    walks a second iterator over the same (unshuffled: src/data_handling/seq2seq_wrappers.py:168-175) dataloader, `window`
    batches at a time; predict_step(i) checks that the batch it was handed holds exactly the tokens that were decoded for
    index i and returns that tensor — identical to generate(batch) (row-scheduled decoding replays the reference's loop per
    given batch).  Any mismatch (a sampler that reorders) switches the look-ahead off for the rest of the run, the counters of
    the batches that were decoded ahead but not served are taken back, and the batch is decoded on the spot.  A batch on which
    the reference raises (or the max_steps guard trips) is left to generate() so that the error surfaces at the same
    predict_step as in the reference — for every generator: the window is decoded with on_error="skip".  Only dataloaders that
    can be iterated again are looked ahead on (a one-shot iterator would be consumed: the module then decodes per batch)."""

    def __init__(self, generator, loader, window: int, in_flight: int):
        self.generator, self.window, self.in_flight = generator, max(1, int(window)), max(1, int(in_flight))
        self.it = iter(loader)
        self.next_idx = 0            # index of the next batch the iterator yields
        self.ready = {}              # batch index -> (src tokens on the device, prediction or None, its counter shares)
        self.enabled = True
        self.served = self.fallbacks = self.windows = 0
        self.decode_seconds = 0.0

    def _decode_window(self, device) -> None:
        pending = []
        for _ in range(self.window):
            try:
                b = next(self.it)
            except StopIteration:
                break
            pending.append(b["src_tokens"].to(device))
        if not pending:
            self.enabled = False
            return
        g = self.generator
        t0 = timer()
        if isinstance(g, D.TranslationInferenceGreedySpeculative):      # slot pools over all rows of the window
            preds = g.generate_many(pending, in_flight=self.in_flight, reorder=True, on_error="skip")
        else:                                                           # source pools over all sources of the window
            preds = g.generate_many(pending, in_flight=self.in_flight, on_error="skip")
        shares = list(getattr(g, "last_batch_counters", [])) or [None] * len(pending)
        self.decode_seconds += timer() - t0
        self.windows += 1
        for k, (src, pred) in enumerate(zip(pending, preds)):
            self.ready[self.next_idx + k] = (src, pred, shares[k])
        self.next_idx += len(pending)

    def _take_back(self, entries) -> None:
        """The generator's counters without the batches that were decoded ahead but will be decoded again by generate()."""
        for _, pred, share in entries:
            if pred is not None and share:
                for name, v in share.items():
                    setattr(self.generator, name, getattr(self.generator, name) - v)

    def take(self, src: torch.Tensor, batch_idx: int):
        """The prediction prepared for batch `batch_idx`, or None (decode it now)."""
        if not self.enabled:
            return None
        if batch_idx not in self.ready and batch_idx == self.next_idx:
            self._decode_window(src.device)
        hit = self.ready.pop(batch_idx, None)
        if hit is None or hit[0].shape != src.shape or not torch.equal(hit[0], src.to(hit[0].device)):
            self.enabled = False                       # not the batch that was decoded for this index: stop looking ahead
            self._take_back(([hit] if hit is not None else []) + list(self.ready.values()))
            self.ready.clear()
            self.fallbacks += 1
            return None
        if hit[1] is None:                             # the reference raises on this batch: let generate() raise it here
            self.fallbacks += 1
            return None
        self.served += 1
        return hit[1]


class VanillaEncoderDecoderTransformerLightning(LightningModule):
    def __init__(self,
                 src_tokenizer=None, tgt_tokenizer=None,
                 embedding_dim: int = 128, feedforward_dim: int = 256, num_encoder_layers: int = 3,
                 num_decoder_layers: int = 3, num_heads: int = 4, dropout_rate: float = 0.0, activation: str = "relu",
                 share_embeddings: bool = False,
                 learning_rate: float = 3e-4, weight_decay: float = 0., scheduler: str = "const", warmup_steps: int = 0,
                 generation: str = "beam_search", beam_size: int = 0, max_len: int = 0, n_drafts: int = 0,
                 draft_len: int = 0, smart_drafts_mode: bool = True,
                 report_prediction_time: bool = True, report_prediction_file: str | None = None):
        super().__init__()
        self.save_hyperparameters(ignore=["src_tokenizer", "tgt_tokenizer"])
        assert src_tokenizer is not None, "source tokenizer not provided"
        assert tgt_tokenizer is not None, "target tokenizer not provided"
        assert activation == "relu", "the HIP path implements the reference configs' ReLU feed-forward"
        self.src_tokenizer, self.tgt_tokenizer = src_tokenizer, tgt_tokenizer
        self.src_vocab_size, self.tgt_vocab_size = src_tokenizer.n_tokens, tgt_tokenizer.n_tokens
        self.src_pad_token_i, self.src_bos_token_i, self.src_eos_token_i = (
            src_tokenizer.pad_token_idx, src_tokenizer.bos_token_idx, src_tokenizer.eos_token_idx)
        self.tgt_pad_token_i, self.tgt_bos_token_i, self.tgt_eos_token_i = (
            tgt_tokenizer.pad_token_idx, tgt_tokenizer.bos_token_idx, tgt_tokenizer.eos_token_idx)
        self.model = WeightContainer(self.src_vocab_size, self.tgt_vocab_size, num_encoder_layers, num_decoder_layers,
                                     embedding_dim, num_heads, feedforward_dim, share_embeddings,
                                     self.src_pad_token_i, self.tgt_pad_token_i)
        if generation not in ("greedy", "beam_search", "greedy_speculative", "beam_search_speculative"):
            options = ", ".join(["beam_search", "greedy", "greedy_speculative", "beam_search_speculative"])
            raise ValueError(f'Unknown generation option {generation}. Options are {options}.')
        if generation == "greedy_speculative":
            assert draft_len > 0, "Number of speculative tokens must be a positive integer."
        self.native: NativeTransformer | None = None
        self.generator = None
        self._ahead = None
        # batches decoded ahead of predict_step (0: decode every batch inside its own predict_step like the reference);
        # not an init_arg, so the reference's YAML files load unchanged — set the attribute or TTX_PREDICT_WINDOW
        self.predict_window = 256
        self.report_prediction_time = report_prediction_time
        self.prediction_start_time = None

    # -- native path ------------------------------------------------------------------------------
    def _weights_fingerprint(self) -> tuple:
        """Content checksum of the parameters (sum and absolute sum per tensor, float64, computed where the weights live): also
        sees writes through ``p.data`` / optimiser swaps that bump no version counter."""
        ps = list(self.model.parameters())
        if not ps:
            return ()
        with torch.no_grad():
            sums = torch.stack([torch.stack((p.detach().double().sum(), p.detach().double().abs().sum())) for p in ps]).cpu()
        return tuple((tuple(p.shape), float(a), float(b)) for p, (a, b) in zip(ps, sums.tolist()))

    def build_native(self, device: int | str | torch.device | None = None, force: bool = False) -> None:
        """(Re)build the HIP model + generator from the current parameters (call after loading a checkpoint).  A second
        call with unchanged parameter VALUES (content checksum) keeps the HIP model and its warm sessions and only makes a
        fresh generator (zeroed counters); ``force=True`` rebuilds regardless."""
        fp = self._weights_fingerprint()
        if not force and self.native is not None and fp == getattr(self, "_native_fp", None) and device is None:
            self.generator = self._create_generator()
            print(self.generator)
            return
        if device is None:
            p = next(self.model.parameters())
            device = p.device if p.is_cuda else torch.device("cuda:0")
        if self.src_pad_token_i != self.tgt_pad_token_i:
            # the reference keeps the two apart (modules.py:44-47); ttx_config carries one pad id, which masks source keys
            # AND target keys, so a source tokenizer with another pad index would be masked wrongly: refuse it loudly
            raise ValueError(f"source pad id {self.src_pad_token_i} != target pad id {self.tgt_pad_token_i}: the HIP path "
                             "supports one shared pad index (the reference's tokenizers fix PAD=0, tokenizer_base.py:27)")
        self.native = NativeTransformer(self.model.state_dict(), self.hparams.num_heads, self.tgt_pad_token_i, device=device)
        self._native_fp = fp
        self.generator = self._create_generator()
        print(self.generator)

    def _create_generator(self):
        h, m = self.hparams, self.native
        common = dict(pad_token=self.tgt_pad_token_i, bos_token=self.tgt_bos_token_i, eos_token=self.tgt_eos_token_i)
        if h.generation == "greedy":
            return D.TranslationInferenceGreedy(m, max_len=h.max_len, **common)
        if h.generation == "beam_search":
            return D.TranslationInferenceBeamSearch(m, beam_size=h.beam_size, max_len=h.max_len, **common)
        if h.generation == "greedy_speculative":
            return D.TranslationInferenceGreedySpeculative(m, max_len=h.max_len, draft_len=h.draft_len, n_drafts=h.n_drafts,
                                                           replace_token=self.tgt_tokenizer.encoder_dict["c"], **common)
        return D.TranslationInferenceBeamSearchSpeculative(
            m, vocab_size=self.tgt_vocab_size, max_len=h.max_len, n_best=h.beam_size, draft_len=h.draft_len,
            n_drafts=h.n_drafts, C_token=self.tgt_tokenizer.encoder_dict["c"], smart_drafts_mode=h.smart_drafts_mode, **common)

    # -- hooks kept from the reference ---------------------------------------------------------------
    def predict_step(self, batch: Any, batch_idx: int, dataloader_idx: int = 0) -> Any:
        if self.generator is None:
            self.build_native()
        ahead = self._ahead
        if ahead is not None and dataloader_idx == 0:
            pred = ahead.take(batch["src_tokens"], batch_idx)
            if pred is not None:
                return pred
        return self.generator.generate(batch["src_tokens"])

    def _predict_loader(self):
        """The (first) predict dataloader Trainer.predict iterates, or None."""
        tr = self.trainer
        if tr is None:
            return None
        loaders = getattr(tr, "predict_dataloaders", None)
        if loaders is None:
            dm = getattr(tr, "datamodule", None)
            loaders = dm.predict_dataloader() if dm is not None and hasattr(dm, "predict_dataloader") else None
        if loaders is None:
            return None
        if isinstance(loaders, (list, tuple)):
            if len(loaders) == 0:
                return None
            return loaders if isinstance(loaders[0], dict) else loaders[0]     # a list of batches is a loader itself
        return loaders

    def on_predict_start(self) -> None:
        self.build_native()                      # weights are final here (Trainer.predict has loaded --ckpt_path)
        self._ahead = None
        window = int(os.environ.get("TTX_PREDICT_WINDOW", str(self.predict_window)))
        if window > 0 and hasattr(self.generator, "generate_many"):
            loader = self._predict_loader()
            # a second pass over the loader must not consume it: one-shot iterators (iter(x) is x) are decoded per batch
            if loader is not None and iter(loader) is not loader:
                self._ahead = _PredictAhead(self.generator, loader, window, int(os.environ.get("TTX_INFLIGHT", "8")))
        if self.report_prediction_time:
            self.prediction_start_time = timer()

    def on_predict_end(self) -> None:
        if not self.report_prediction_time:
            return
        torch.cuda.synchronize()
        elapsed = datetime.timedelta(seconds=timer() - self.prediction_start_time)
        h = self.hparams
        dm = getattr(self.trainer, "datamodule", None)
        report = {
            "algorithm": h.generation,
            "batch_size": getattr(dm, "batch_size", None),
            "tgt_test_path": str(getattr(dm, "tgt_test_path", None)),
            "max_len": h.max_len,
            "total_seconds": round(elapsed.total_seconds(), 4),
            "model_calls": self.generator.model_calls_num,
            "seconds_per_model_call": round(elapsed.total_seconds() / max(1, self.generator.model_calls_num), 4),
        }
        if h.generation in ("greedy_speculative", "beam_search_speculative"):
            report["n_drafts"] = h.n_drafts
            report["draft_len"] = h.draft_len
            if h.generation == "beam_search_speculative":
                report["accepted_tokens"] = self.generator.accepted_tokens_num
                report["acceptance_rate"] = round(self.generator.accepted_tokens_num /
                                                  max(1, self.generator.produced_non_pad_tokens), 4)
        text = json.dumps(report)
        print(text)
        if h.report_prediction_file is not None:
            Path(h.report_prediction_file).parent.mkdir(exist_ok=True)
            with open(h.report_prediction_file, "a") as f:
                print(text, file=f)

    # -- out of scope ------------------------------------------------------------
    def training_step(self, *a, **k):
        raise NotImplementedError("training is outside the MI355X inference path; train with the reference")

    validation_step = test_step = training_step


def run_predict(module: VanillaEncoderDecoderTransformerLightning, batches, writer=None, datamodule=None,
                schedule: str = "rows", window: int = 256, in_flight: int = 8) -> list:
    """Stand-in for ``Trainer.predict`` when pytorch_lightning is absent: the same hook order and nothing else
    (on_predict_start -> predict_step per batch -> writer.write_on_batch_end -> on_predict_end), with a trainer object that
    exposes ``predict_dataloaders`` and ``datamodule`` like Lightning's.  ``schedule="batches"`` switches the module's
    look-ahead off (every batch is decoded inside its own predict_step, as in the reference)."""
    if schedule not in ("batches", "rows"):
        raise ValueError("schedule must be 'batches' or 'rows'")
    batches = list(batches)
    module.trainer = SimpleNamespace(datamodule=datamodule, predict_dataloaders=batches)
    module.predict_window = window if schedule == "rows" else 0
    old_inflight = os.environ.get("TTX_INFLIGHT")
    os.environ["TTX_INFLIGHT"] = str(in_flight)
    outs = []
    try:
        with torch.inference_mode():
            module.on_predict_start()
            for i, batch in enumerate(batches):
                pred = module.predict_step(batch, i)
                if writer is not None:
                    writer.write_on_batch_end(module.trainer, module, pred, None, batch, i, 0)
                outs.append(pred)
            module.on_predict_end()
    finally:
        if old_inflight is None:
            os.environ.pop("TTX_INFLIGHT", None)
        else:
            os.environ["TTX_INFLIGHT"] = old_inflight
    return outs
