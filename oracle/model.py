"""ORACLE (test infrastructure, never shipped, never measured as the product).

CPU restatement, in plain fp32 torch tensor algebra, of the reference's encoder-decoder forward:

  * VanillaTransformer.encode_src   src/model/modules.py:110-116
  * VanillaTransformer.decode_tgt   src/model/modules.py:118-138
  * VanillaTransformer.forward      src/model/modules.py:86-108
  * TokenEmbedding / PositionalEncoding   src/model/embeddings.py:8-15, :30-64

The reference builds these from stock ``torch.nn.Transformer`` modules (post-norm, ReLU, eps 1e-5,
batch_first; src/model/modules.py:52-81).  This file spells the same arithmetic out by hand
(matmul / softmax / layer-norm on explicit weight tensors) so that every intermediate is
addressable for stage-by-stage comparison with the HIP kernels, and so that it does not depend on
which fused fast path a given torch build picks.  Weights use the reference's state-dict names
(SURVEY.md §8(b) row B6), with or without the Lightning ``model.`` prefix.

Pinned by: tests/golden/tiny_model_io.npz and tests/golden/full_model_io.npz, both produced by
importing the reference itself (tests/golden/make_golden.py); checked in tests/test_oracle_model.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

NEG_INF = float("-inf")


@dataclass
class OracleConfig:
    vocab_size: int
    embedding_dim: int = 256
    num_heads: int = 8
    feedforward_dim: int = 2048
    num_encoder_layers: int = 4
    num_decoder_layers: int = 4
    pad_token_idx: int = 0
    layer_norm_eps: float = 1e-5
    max_positions: int = 5000


def positional_table(emb: int, max_len: int = 5000) -> torch.Tensor:
    """Row 0 is all zeros; row p+1 holds the sin/cos code of position p (embeddings.py:38-45)."""
    pos = torch.arange(0, max_len, dtype=torch.float32).unsqueeze(1)
    freq = torch.exp(torch.arange(0, emb, 2, dtype=torch.float32) * (-math.log(10000.0) / emb))
    table = torch.zeros(max_len + 1, emb, dtype=torch.float32)
    table[1:, 0::2] = torch.sin(pos * freq)
    table[1:, 1::2] = torch.cos(pos * freq)
    return table


def strip_prefix(state: dict, prefix: str = "model.") -> dict:
    if any(k.startswith(prefix) for k in state):
        return {k[len(prefix):]: v for k, v in state.items() if k.startswith(prefix)}
    return dict(state)


class OracleTransformer:
    """Functional fp32 model over a reference-layout state dict.  Exposes the L3->L2 protocol of
    SURVEY.md §8(b) B5: ``src_pad_token_i``, ``encode_src``, ``decode_tgt``, ``__call__(src, tgt)``."""

    def __init__(self, cfg: OracleConfig, state: dict, device: str | torch.device = "cpu"):
        self.cfg = cfg
        self.device = torch.device(device)
        st = strip_prefix(state)
        self.w = {k: torch.as_tensor(v, dtype=torch.float32).to(self.device) for k, v in st.items()}
        self.src_pad_token_i = cfg.pad_token_idx
        self.tgt_pad_token_i = cfg.pad_token_idx
        self.pe = positional_table(cfg.embedding_dim, cfg.max_positions).to(self.device)
        self.trace: dict | None = None  # set to {} to record intermediates

    # -- helpers ------------------------------------------------------------------------
    def _rec(self, name: str, value: torch.Tensor) -> None:
        if self.trace is not None:
            self.trace[name] = value.detach().clone()

    def _embed(self, tokens: torch.Tensor, which: str) -> torch.Tensor:
        # embeddings.py:14-15 (no sqrt(d) scaling) + :60-64 with offset 0 -> pe rows 1..L
        table = self.w[f"{which}_token_featurizer.embedding.weight"]
        L = tokens.size(1)
        return F.embedding(tokens, table) + self.pe[1:L + 1].unsqueeze(0)

    def _mha(self, prefix: str, x_q: torch.Tensor, x_kv: torch.Tensor, add_mask: torch.Tensor | None):
        """torch.nn.MultiheadAttention arithmetic: packed in-proj rows are Q, K, V in that order;
        scores scaled by 1/sqrt(dh); ``add_mask`` is an additive float mask broadcastable to
        [B, H, Lq, Lk]."""
        cfg = self.cfg
        E, H = cfg.embedding_dim, cfg.num_heads
        dh = E // H
        w_in, b_in = self.w[prefix + ".in_proj_weight"], self.w[prefix + ".in_proj_bias"]
        q = x_q @ w_in[:E].T + b_in[:E]
        k = x_kv @ w_in[E:2 * E].T + b_in[E:2 * E]
        v = x_kv @ w_in[2 * E:].T + b_in[2 * E:]
        B, Lq, _ = q.shape
        Lk = k.size(1)
        q = q.view(B, Lq, H, dh).transpose(1, 2)
        k = k.view(B, Lk, H, dh).transpose(1, 2)
        v = v.view(B, Lk, H, dh).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
        if add_mask is not None:
            s = s + add_mask
        p = torch.softmax(s, dim=-1)
        o = (p @ v).transpose(1, 2).reshape(B, Lq, E)
        return o @ self.w[prefix + ".out_proj.weight"].T + self.w[prefix + ".out_proj.bias"]

    def _ln(self, prefix: str, x: torch.Tensor) -> torch.Tensor:
        return F.layer_norm(x, (self.cfg.embedding_dim,), self.w[prefix + ".weight"], self.w[prefix + ".bias"],
                            self.cfg.layer_norm_eps)

    def _ffn(self, prefix: str, x: torch.Tensor) -> torch.Tensor:
        h = torch.relu(x @ self.w[prefix + ".linear1.weight"].T + self.w[prefix + ".linear1.bias"])
        return h @ self.w[prefix + ".linear2.weight"].T + self.w[prefix + ".linear2.bias"]

    # -- encoder ------------------------------------------------------------------------
    def encode_src(self, src: torch.Tensor, src_pad_mask: torch.Tensor) -> torch.Tensor:
        """modules.py:110-116.  ``src_pad_mask`` True = PAD key.  Rows of the result at PAD positions are
        set to zero: in eval/inference mode the reference's TransformerEncoder takes the nested-tensor
        fast path, which zero-fills them; they are masked by every consumer anyway."""
        x = self._embed(src, "src")
        key_mask = torch.zeros(src_pad_mask.shape, dtype=torch.float32, device=x.device)
        key_mask = key_mask.masked_fill(src_pad_mask, NEG_INF)[:, None, None, :]
        self._rec("enc.embed", x)
        for i in range(self.cfg.num_encoder_layers):
            p = f"transformer.encoder.layers.{i}"
            x = self._ln(p + ".norm1", x + self._mha(p + ".self_attn", x, x, key_mask))
            x = self._ln(p + ".norm2", x + self._ffn(p, x))
            self._rec(f"enc.layer{i}", x)
        x = self._ln("transformer.encoder.norm", x)
        return x.masked_fill(src_pad_mask.unsqueeze(-1), 0.0)

    # -- decoder ------------------------------------------------------------------------
    def decode_hidden(self, tgt: torch.Tensor, memory: torch.Tensor, memory_pad_mask: torch.Tensor) -> torch.Tensor:
        x = self._embed(tgt, "tgt")
        Lt = tgt.size(1)
        causal = torch.full((Lt, Lt), NEG_INF, device=x.device).triu(1)  # modules.py:128
        self_mask = causal[None, None] + torch.zeros(tgt.shape, dtype=torch.float32, device=x.device) \
            .masked_fill(tgt == self.tgt_pad_token_i, NEG_INF)[:, None, None, :]  # modules.py:127
        mem_mask = torch.zeros(memory_pad_mask.shape, dtype=torch.float32, device=x.device) \
            .masked_fill(memory_pad_mask, NEG_INF)[:, None, None, :]
        self._rec("dec.embed", x)
        for i in range(self.cfg.num_decoder_layers):
            p = f"transformer.decoder.layers.{i}"
            x = self._ln(p + ".norm1", x + self._mha(p + ".self_attn", x, x, self_mask))
            self._rec(f"dec.layer{i}.sa", x)
            x = self._ln(p + ".norm2", x + self._mha(p + ".multihead_attn", x, memory, mem_mask))
            self._rec(f"dec.layer{i}.ca", x)
            x = self._ln(p + ".norm3", x + self._ffn(p, x))
            self._rec(f"dec.layer{i}", x)
        return self._ln("transformer.decoder.norm", x)

    def decode_tgt(self, tgt: torch.Tensor, memory: torch.Tensor, memory_pad_mask: torch.Tensor) -> torch.Tensor:
        """modules.py:118-138: full-prefix decoder forward + classifier, logits for every position."""
        h = self.decode_hidden(tgt, memory, memory_pad_mask)
        return h @ self.w["next_token_classifier.weight"].T + self.w["next_token_classifier.bias"]

    def __call__(self, src: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
        """modules.py:86-108 (used by step 0 of standard beam search, standard_decoding.py:102)."""
        mask = src == self.src_pad_token_i
        memory = self.encode_src(src, mask)
        return self.decode_tgt(tgt, memory, mask)


def config_from_state(state: dict, num_heads: int, pad_token_idx: int = 0) -> OracleConfig:
    st = strip_prefix(state)
    emb = st["src_token_featurizer.embedding.weight"]
    n_enc = 1 + max(int(k.split(".")[3]) for k in st if k.startswith("transformer.encoder.layers."))
    n_dec = 1 + max(int(k.split(".")[3]) for k in st if k.startswith("transformer.decoder.layers."))
    return OracleConfig(vocab_size=int(st["next_token_classifier.weight"].shape[0]),
                        embedding_dim=int(emb.shape[1]), num_heads=num_heads,
                        feedforward_dim=int(st["transformer.encoder.layers.0.linear1.weight"].shape[0]),
                        num_encoder_layers=n_enc, num_decoder_layers=n_dec, pad_token_idx=pad_token_idx)
