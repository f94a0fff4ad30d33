"""Host-side mirror of the reference tokenizer classes over the C++ scanner in libttx_hip.so
(``ChemSMILESTokenizer`` / ``GenericTokenizer``: src/data_handling/tokenizer_smiles.py, tokenizer_base.py).
Same attribute and method names, so it can be handed to the Lightning module and the PredictionWriter in
place of the reference tokenizer.  No GPU involved."""
from __future__ import annotations

import ctypes as C
import json
from pathlib import Path
from typing import Iterable

import numpy as np

from . import _native as N

BOS_TOKEN, EOS_TOKEN, PAD_TOKEN, UNK_TOKEN = "<BOS>", "<EOS>", "<PAD>", "?"


class NativeSmilesTokenizer:
    pad_token_idx, bos_token_idx, eos_token_idx, unk_token_idx = 0, 1, 2, 3

    def __init__(self, vocab: dict | None = None):
        self.bos_token, self.eos_token, self.pad_token, self.unk_token = BOS_TOKEN, EOS_TOKEN, PAD_TOKEN, UNK_TOKEN
        self._h = None
        self.decoder_dict = {0: PAD_TOKEN, 1: BOS_TOKEN, 2: EOS_TOKEN, 3: UNK_TOKEN}
        self.encoder_dict = {v: k for k, v in self.decoder_dict.items()}
        if vocab is not None:
            self.assign_vocab(vocab)
        else:
            self._rebuild()

    # -- vocabulary ------------------------------------------------------------------------------
    @property
    def n_tokens(self) -> int:
        return len(self.encoder_dict)

    def load_vocab(self, voc_load_path) -> None:
        p = Path(voc_load_path).resolve()
        if not p.exists():
            raise FileNotFoundError
        self.decoder_dict = {int(k): v for k, v in json.loads(p.read_text()).items()}
        self.encoder_dict = {v: k for k, v in self.decoder_dict.items()}
        self._rebuild()

    def assign_vocab(self, vocab: dict) -> None:
        self.encoder_dict = dict(vocab)
        self.decoder_dict = {v: k for k, v in vocab.items()}
        self._rebuild()

    def save_vocab(self, voc_save_path) -> None:
        p = Path(voc_save_path).resolve()
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(json.dumps(self.decoder_dict, sort_keys=True))

    def _rebuild(self) -> None:
        lib = N.lib()
        if self._h:
            lib.ttx_tokenizer_destroy(self._h)
        items = sorted(self.decoder_dict.items())
        toks = (C.c_char_p * len(items))(*[v.encode("utf-8") for _, v in items])
        ids = (C.c_int32 * len(items))(*[k for k, _ in items])
        h = C.c_void_p()
        N.check(lib.ttx_tokenizer_create(toks, ids, len(items), C.byref(h)))
        self._h = h

    def __del__(self):
        try:
            if self._h:
                N.lib().ttx_tokenizer_destroy(self._h)
        except Exception:
            pass

    # -- strings <-> ids ---------------------------------------------------------------------------
    def encode(self, seq: str) -> list[int]:
        lib = N.lib()
        raw = seq.encode("utf-8")
        cap = len(raw) + 2
        buf = (C.c_int32 * cap)()
        n = lib.ttx_tokenizer_encode(self._h, raw, buf, cap)
        if n < 0:
            N.check(n)
        return list(buf[:n])

    def encode_batch(self, lines: list[str]) -> np.ndarray:
        """Tokenize and pad (what Seq2SeqDataset + collate_fn produce for one batch): int64 [B, Lmax]."""
        lib = N.lib()
        raws = [l.encode("utf-8") for l in lines]
        cap = max(len(r) for r in raws) + 2
        out = np.empty((len(lines), cap), dtype=np.int64)
        arr = (C.c_char_p * len(raws))(*raws)
        w = lib.ttx_tokenizer_encode_batch(self._h, arr, len(raws), out.ctypes.data, cap)
        if w <= 0:                      # cannot happen with cap = longest line + 2; report it loudly if it does
            raise RuntimeError(f"ttx_tokenizer_encode_batch needs {-w} columns, {cap} given")
        return np.ascontiguousarray(out[:, :w])

    def decode(self, tokens: Iterable[int], skip_service_tokens: bool = True) -> str:
        if not skip_service_tokens:
            return "".join(self.decoder_dict[int(i)] for i in tokens)
        ids = np.ascontiguousarray(np.asarray(list(tokens) if not isinstance(tokens, np.ndarray) else tokens, dtype=np.int64))
        lib = N.lib()
        cap = 16 * max(1, ids.size) + 1
        buf = C.create_string_buffer(cap)
        n = lib.ttx_tokenizer_decode(self._h, ids.ctypes.data, int(ids.size), buf, cap)
        if n < 0:
            raise KeyError(lib.ttx_last_error().decode())
        if n >= cap:
            buf = C.create_string_buffer(n + 1)
            lib.ttx_tokenizer_decode(self._h, ids.ctypes.data, int(ids.size), buf, n + 1)
        return buf.value.decode("utf-8")

    def decode_batch(self, tokens) -> list[str]:
        return [self.decode(i) for i in tokens]
