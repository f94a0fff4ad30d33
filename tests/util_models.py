"""Shared helpers for the tests: golden loaders, the seeded full-size weights, token trimming."""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import torch

GOLDEN = Path(__file__).resolve().parent / "golden"
PAD, BOS, EOS = 0, 1, 2


def load_npz(name: str) -> dict:
    z = np.load(GOLDEN / name)
    return {k: z[k] for k in z.files}


def fixture_tokens():
    z = load_npz("fixture_tokens.npz")
    return torch.from_numpy(z["src"]), torch.from_numpy(z["tgt"]), int(z["c_token"]), int(z["vocab_size"])


def tiny_state() -> tuple[dict, dict]:
    cfg = json.loads((GOLDEN / "tiny_config.json").read_text())
    return load_npz("tiny_weights.npz"), cfg


def seeded_weights(shapes, seed: int) -> dict:
    """Same rule as tests/golden/make_golden.py:seeded_weights (splitmix64 -> uniform); regenerates the
    full-size weights the reference outputs in full_model_io.npz were computed with."""
    out = {}
    mask64 = (1 << 64) - 1
    ctr = np.uint64(seed)
    with np.errstate(over="ignore"):
        for name, shape in shapes:
            n = int(np.prod(shape))
            z = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) + ctr
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
            u = (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)
            if name.endswith(("norm1.weight", "norm2.weight", "norm3.weight", "norm.weight")):
                w = 1.0 + 0.2 * (u - 0.5)
            elif len(shape) == 1:
                w = 0.2 * (u - 0.5)
            else:
                w = (1.7 / np.sqrt(shape[-1])) * (2.0 * u - 1.0)
            out[name] = w.astype(np.float32).reshape(shape)
            ctr = np.uint64((int(ctr) + 0x632BE59BD9B4E019 * (n + 1)) & mask64)
    return out


def state_shapes(V: int, d: int, F: int, n_enc: int, n_dec: int):
    """Reference state-dict names and shapes in registration order."""
    s = [("src_token_featurizer.embedding.weight", (V, d)), ("tgt_token_featurizer.embedding.weight", (V, d))]

    def attn(p):
        return [(p + ".in_proj_weight", (3 * d, d)), (p + ".in_proj_bias", (3 * d,)),
                (p + ".out_proj.weight", (d, d)), (p + ".out_proj.bias", (d,))]

    def ffn(p):
        return [(p + ".linear1.weight", (F, d)), (p + ".linear1.bias", (F,)),
                (p + ".linear2.weight", (d, F)), (p + ".linear2.bias", (d,))]

    def norm(p):
        return [(p + ".weight", (d,)), (p + ".bias", (d,))]

    for i in range(n_enc):
        p = f"transformer.encoder.layers.{i}"
        s += attn(p + ".self_attn") + ffn(p) + norm(p + ".norm1") + norm(p + ".norm2")
    s += norm("transformer.encoder.norm")
    for i in range(n_dec):
        p = f"transformer.decoder.layers.{i}"
        s += attn(p + ".self_attn") + attn(p + ".multihead_attn") + ffn(p)
        s += norm(p + ".norm1") + norm(p + ".norm2") + norm(p + ".norm3")
    s += norm("transformer.decoder.norm")
    s += [("next_token_classifier.weight", (V, d)), ("next_token_classifier.bias", (V,))]
    return s


def full_state(V: int = 64, seed: int = 20250725) -> dict:
    w = seeded_weights(state_shapes(V, 256, 2048, 4, 4), seed)
    w["tgt_token_featurizer.embedding.weight"] = w["src_token_featurizer.embedding.weight"]
    return w


def upto_eos(row) -> list[int]:
    """Tokens up to and including the first EOS (parity is defined on these: SURVEY.md §8(a) quirk 4)."""
    out = []
    for t in np.asarray(row).tolist():
        out.append(int(t))
        if t == EOS:
            break
    else:
        while out and out[-1] == PAD:
            out.pop()
    return out
