"""Host-side mirror of the reference model protocol over the HIP library.

``NativeTransformer`` offers what the reference's generators call on ``VanillaTransformer``
(src/model/modules.py:86-138): ``src_pad_token_i``, ``encode_src(src, src_pad_mask)``,
``decode_tgt(tgt, memory, memory_pad_mask=...)`` and ``model(src, tgt)`` — same argument meaning, same
tensor shapes/dtypes out — so the reference's own generator classes can drive it unchanged, and it owns
the ``ttx_session`` the native generators of decoding.py run on.  torch is used for device memory and
the current stream only; every FLOP happens in libttx_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os
import math

import numpy as np
import torch

from . import _native as N


def reference_pe_table(emb: int, max_len: int = 5000) -> torch.Tensor:
    """The non-persistent ``pe`` buffer of the reference (src/model/embeddings.py:38-45), rebuilt with the
    same torch ops so the table handed to the library is bit-identical to the one the reference adds."""
    pe = torch.zeros(max_len, emb)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, emb, 2).float() * (-math.log(10000.0) / emb))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return torch.vstack((torch.zeros(1, emb), pe)).contiguous()


def _strip(state: dict) -> dict:
    if any(k.startswith("model.") for k in state):
        return {k[len("model."):]: v for k, v in state.items() if k.startswith("model.")}
    return dict(state)


def shape_of_state(state_dict: dict) -> dict:
    """Model dimensions read off a reference state dict (SURVEY.md §8(b) B6 names)."""
    st = _strip(state_dict)
    emb = st["src_token_featurizer.embedding.weight"]
    return {"emb_dim": int(emb.shape[1]), "src_vocab_size": int(emb.shape[0]),
            "tgt_vocab_size": int(st["next_token_classifier.weight"].shape[0]),
            "ff_dim": int(st["transformer.encoder.layers.0.linear1.weight"].shape[0]),
            "num_enc_layers": 1 + max(int(k.split(".")[3]) for k in st if k.startswith("transformer.encoder.layers.")),
            "num_dec_layers": 1 + max(int(k.split(".")[3]) for k in st if k.startswith("transformer.decoder.layers."))}


class _DeviceSpan:
    """Exposes a device allocation of the library to torch (``torch.as_tensor``) without copying it."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes // 4,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class NativeTransformer:
    def __init__(self, state_dict: dict | None, num_heads: int, pad_token_idx: int = 0, device: int | str | torch.device = 0,
                 max_positions: int = 5000, layer_norm_eps: float = 1e-5, shape: dict | None = None):
        """``state_dict``: the reference's state dict (weights are packed into one HBM blob).  ``state_dict=None`` with
        ``shape`` (see shape_of_state): an EMPTY model of those dimensions whose blob is filled afterwards — the receiving side
        of the one-off weight broadcast (dist.broadcast_model)."""
        if not torch.cuda.is_available():
            raise RuntimeError("NativeTransformer needs an MI355X: the HIP path has no CPU fallback")
        dev = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        self.device = dev
        st = _strip(state_dict) if state_dict is not None else None
        dims = shape_of_state(st) if st is not None else dict(shape)
        self.emb_dim, self.src_vocab_size, self.tgt_vocab_size = dims["emb_dim"], dims["src_vocab_size"], dims["tgt_vocab_size"]
        self.ff_dim, self.num_enc_layers, self.num_dec_layers = dims["ff_dim"], dims["num_enc_layers"], dims["num_dec_layers"]
        self.num_heads = int(num_heads)
        self.src_pad_token_i = int(pad_token_idx)
        self.tgt_pad_token_i = int(pad_token_idx)
        self.cfg = N.Config(self.tgt_vocab_size, self.src_vocab_size, self.emb_dim, self.num_heads, self.ff_dim,
                            self.num_enc_layers, self.num_dec_layers, self.src_pad_token_i, int(max_positions),
                            float(layer_norm_eps))
        self._lib = N.lib()
        self._model = C.c_void_p()
        self._session = C.c_void_p()
        # TTX_PROFILE_GEMM=1 at construction makes EVERY session of this model a profiling one (new_session below), not only the
        # first: the slot pools' later sessions are created long after the caller dropped the variable again
        self._profile_sessions = os.environ.get("TTX_PROFILE_GEMM") == "1"
        if st is None:
            N.check(self._lib.ttx_model_create_empty(C.byref(self.cfg), dev.index or 0, C.byref(self._model)))
            N.check(self._lib.ttx_session_create(self._model, C.byref(self._session)))
            return
        host = {k: torch.as_tensor(v).detach().to("cpu", torch.float32).contiguous() for k, v in st.items()}
        host["positional_encoding.pe"] = reference_pe_table(self.emb_dim, max_positions)
        arr = (N.Tensor * len(host))()
        keep = []
        for i, (k, v) in enumerate(host.items()):
            name = k.encode()
            keep.append((name, v))
            arr[i] = N.Tensor(name, C.cast(v.data_ptr(), C.POINTER(C.c_float)), v.numel())
        N.check(self._lib.ttx_model_create(C.byref(self.cfg), arr, len(host), dev.index or 0, C.byref(self._model)))
        N.check(self._lib.ttx_session_create(self._model, C.byref(self._session)))

    def blob_tensor(self) -> torch.Tensor:
        """The packed weight blob in HBM as a flat fp32 torch tensor sharing the library's memory (ttx_model_blob): every
        tensor of the state dict plus the derived ones (sinusoid table, packed cross-attention K/V projection).  Writing
        the blob of another model of the same shape into it (one RCCL broadcast) makes this model that model."""
        ptr, nbytes = C.c_void_p(), C.c_int64()
        N.check(self._lib.ttx_model_blob(self._model, C.byref(ptr), C.byref(nbytes)))
        with torch.cuda.device(self.device):
            t = torch.as_tensor(_DeviceSpan(int(ptr.value), int(nbytes.value)), device=self.device)
        t._ttx_owner = self              # the tensor borrows the allocation: keep the model alive with it
        return t

    # -- plumbing ------------------------------------------------------------------------------
    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @property
    def session(self) -> C.c_void_p:
        return self._session

    def new_session(self) -> C.c_void_p:
        s = C.c_void_p()
        had = os.environ.get("TTX_PROFILE_GEMM")
        if self._profile_sessions:
            os.environ["TTX_PROFILE_GEMM"] = "1"
        try:
            N.check(self._lib.ttx_session_create(self._model, C.byref(s)))
        finally:
            if self._profile_sessions:
                if had is None:
                    os.environ.pop("TTX_PROFILE_GEMM", None)
                else:
                    os.environ["TTX_PROFILE_GEMM"] = had
        return s

    def session_pool(self, n: int) -> list:
        """`n` sessions (the default one first) for several batches in flight; created once, reused."""
        pool = getattr(self, "_pool", None)
        if pool is None:
            pool = self._pool = [self._session]
        while len(pool) < n:
            pool.append(self.new_session())
        return pool[:n]

    def kernel_profile(self) -> dict:
        """GEMM event-pair sums of every session of this model since the last read (ttx_last_kernel_profile; sessions created
        under TTX_PROFILE_GEMM=1): {"gemm_ms", "launches", "pair_overhead_ms"}."""
        ms, n, e = C.c_double(), C.c_int64(), C.c_double()
        tot_ms, tot_n, over = 0.0, 0, []
        for sess in (getattr(self, "_pool", None) or [self._session]):
            N.check(self._lib.ttx_last_kernel_profile(sess, C.byref(ms), C.byref(n), C.byref(e)))
            tot_ms += ms.value
            tot_n += n.value
            if n.value and e.value > 0:
                over.append(e.value)
        return {"gemm_ms": tot_ms, "launches": tot_n, "pair_overhead_ms": float(np.median(over)) if over else 0.0}

    def close(self) -> None:
        for extra in getattr(self, "_pool", [])[1:]:
            self._lib.ttx_session_destroy(extra)
        self._pool = None
        if getattr(self, "_session", None):
            self._lib.ttx_session_destroy(self._session)
            self._session = None
        if getattr(self, "_model", None):
            self._lib.ttx_model_destroy(self._model)
            self._model = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _tokens(self, t: torch.Tensor) -> torch.Tensor:
        return t.to(self.device, torch.int64).contiguous()

    def check_tokens(self, t: torch.Tensor, vocab: int | None = None) -> None:
        """torch.nn.Embedding raises IndexError on ids outside the table (the reference's first op on `src`); so do the
        generators here, before the ids reach a kernel (which would otherwise look them up as id 0)."""
        if t.numel() == 0:
            return
        lo, hi = int(t.min()), int(t.max())
        if lo < 0 or hi >= (vocab or self.src_vocab_size):
            raise IndexError("index out of range in self")

    # -- B5 ------------------------------------------------------------------------------------
    def encode_src(self, src: torch.Tensor, src_pad_mask: torch.Tensor | None = None) -> torch.Tensor:
        """modules.py:110-116.  The mask argument is accepted for signature parity; the library derives
        it as ``src == pad`` exactly as every reference call site does (speculative_decoding.py:60)."""
        src = self._tokens(src)
        self.check_tokens(src)
        B, Ls = src.shape
        mem = torch.empty((B, Ls, self.emb_dim), dtype=torch.float32, device=self.device)
        N.check(self._lib.ttx_encode_src(self._session, src.data_ptr(), B, Ls, mem.data_ptr(), self._stream()))
        return mem

    def decode_tgt(self, tgt: torch.Tensor, memory: torch.Tensor, memory_pad_mask: torch.Tensor,
                   memory_row: torch.Tensor | None = None) -> torch.Tensor:
        """modules.py:118-138.  ``memory_row`` (int32 [R], optional) lets R decoder rows share fewer
        memory rows instead of the reference's repeat_interleave'd copy."""
        tgt = self._tokens(tgt)
        R, Lt = tgt.shape
        memory = memory.to(self.device, torch.float32).contiguous()
        Rm, Ls, _ = memory.shape
        pad = memory_pad_mask.to(self.device, torch.uint8).contiguous()
        row_ptr = None
        if memory_row is not None:
            memory_row = memory_row.to(self.device, torch.int32).contiguous()
            row_ptr = memory_row.data_ptr()
        logits = torch.empty((R, Lt, self.tgt_vocab_size), dtype=torch.float32, device=self.device)
        N.check(self._lib.ttx_decode_tgt(self._session, tgt.data_ptr(), R, Lt, memory.data_ptr(), pad.data_ptr(),
                                         row_ptr, Rm, Ls, logits.data_ptr(), self._stream()))
        return logits

    def __call__(self, src: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
        """modules.py:86-108."""
        src, tgt = self._tokens(src), self._tokens(tgt)
        B, Ls = src.shape
        Lt = tgt.shape[1]
        logits = torch.empty((B, Lt, self.tgt_vocab_size), dtype=torch.float32, device=self.device)
        N.check(self._lib.ttx_forward(self._session, src.data_ptr(), B, Ls, tgt.data_ptr(), Lt, logits.data_ptr(),
                                      self._stream()))
        return logits

    forward = __call__

    # -- beam-speculative bookkeeping kernels --------------------------------------------------
    def nucleus_mask(self, logits: torch.Tensor, nucleus: float, max_kept: int, fill: float) -> torch.Tensor:
        """mask_with_num_logits_according_nucleus (speculative_decoding.py:871-904) on the device."""
        shape = logits.shape
        x = logits.to(self.device, torch.float32).contiguous().reshape(-1, shape[-1])
        out = torch.empty_like(x)
        N.check(self._lib.ttx_nucleus_mask(self._session, x.data_ptr(), x.shape[0], x.shape[1], float(nucleus), int(max_kept),
                                           float(fill), out.data_ptr(), self._stream()))
        return out.reshape(shape)

    def accepted_lengths(self, logits: torch.Tensor, drafts: torch.Tensor, nucleus: float, max_kept: int) -> torch.Tensor:
        """Leading draft tokens that survive the nucleus mask of their position (speculative_decoding.py:539-548, :847-869).
        logits [R,D+1,V], drafts Long[R,D] -> Long[R]."""
        R, D1, V = logits.shape
        x = logits.to(self.device, torch.float32).contiguous()
        d = drafts.to(self.device, torch.int64).contiguous()
        out = torch.empty(R, dtype=torch.int32, device=self.device)
        N.check(self._lib.ttx_accepted_lengths(self._session, x.data_ptr(), d.data_ptr(), R, D1 - 1, V, float(nucleus),
                                               int(max_kept), out.data_ptr(), self._stream()))
        return out.long()

    def ragged_topk(self, score: torch.Tensor, counts: torch.Tensor, k: int):
        """topk_in_each_group (speculative_decoding.py:177-238): (values [G,k], flat indices [G*k])."""
        sc = score.to(self.device, torch.float32).contiguous().reshape(-1)
        counts = counts.to(self.device)
        offs = torch.zeros(counts.numel() + 1, dtype=torch.int32, device=self.device)
        offs[1:] = counts.cumsum(0)
        G = counts.numel()
        top = torch.empty((G, k), dtype=torch.float32, device=self.device)
        idx = torch.empty((G, k), dtype=torch.int64, device=self.device)
        N.check(self._lib.ttx_ragged_topk(self._session, sc.data_ptr(), offs.data_ptr(), G, int(counts.max()), int(k),
                                          top.data_ptr(), idx.data_ptr(), self._stream()))
        return top, idx.reshape(-1)

    def make_drafts(self, src: torch.Tensor, draft_len: int, n_drafts: int, min_draft_len: int, max_draft_len: int,
                    eos_token_idx: int, pad_token_idx: int, replace_token_idx: int) -> torch.Tensor:
        """src/utils/drafting.py:5-67 on the device."""
        src = self._tokens(src)
        B, L = src.shape
        D = min(max(min_draft_len, draft_len), max_draft_len)
        out = torch.empty((B, n_drafts, D), dtype=torch.int64, device=self.device)
        N.check(self._lib.ttx_make_drafts(self._session, src.data_ptr(), B, L, draft_len, n_drafts, min_draft_len,
                                          max_draft_len, eos_token_idx, pad_token_idx, replace_token_idx,
                                          out.data_ptr(), self._stream()))
        return out
