#!/usr/bin/env python3
"""Would a split-bf16 FFN change any token?  CPU experiment for DESIGN.md §8 item 1: the oracle model with its two FFN products
computed from bf16 pieces (x = hi + mid + lo, bf16 x bf16 products exact in fp32, fp32 accumulation per partial product — what
six (or three) bf16 MFMAs per fp32 MFMA would compute), run through every golden generator case; reports how many golden
hypotheses change.  Test infrastructure only (imports oracle/).  Usage: python tests/tools/bf16_split_tokens.py [3|6]"""
import sys
from pathlib import Path


import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from oracle.model import OracleTransformer, config_from_state  # noqa: E402
from oracle.decoding import GreedyOracle, BeamSearchOracle  # noqa: E402
from oracle.spec_beam import BeamSearchSpeculativeOracle  # noqa: E402
from util_models import load_npz, fixture_tokens, tiny_state, upto_eos, PAD, BOS, EOS  # noqa: E402

TERMS = {3: (2, [(0, 0), (0, 1), (1, 0)]), 6: (3, [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)])}
parts, terms = TERMS[int(sys.argv[1]) if len(sys.argv) > 1 else 6]


def split(t):
    out, r = [], t.float()
    for _ in range(parts):
        h = r.bfloat16().float()
        out.append(h)
        r = r - h
    return out


def mm(a, w):
    A, W = split(a), split(w)
    acc = torch.zeros(a.shape[:-1] + (w.shape[0],), dtype=torch.float32)
    for i, j in terms:
        acc = acc + A[i] @ W[j].T
    return acc


class SplitFFN(OracleTransformer):
    def _ffn(self, prefix, x):
        h = torch.relu(mm(x, self.w[prefix + ".linear1.weight"]) + self.w[prefix + ".linear1.bias"])
        return mm(h, self.w[prefix + ".linear2.weight"]) + self.w[prefix + ".linear2.bias"]


st, cfg = tiny_state()
model = SplitFFN(config_from_state(st, cfg["num_heads"]), st)
src, _, c, V = fixture_tokens()
same = total = 0
gold = load_npz("gen_greedy.npz")
for bsz in (1, 4, 10):
    g = GreedyOracle(model, 150, PAD, BOS, EOS)
    for i in range(0, 10, bsz):
        out = g.generate(src[i:i + bsz]).numpy()
        for a, b in zip(out[:, 0], gold[f"b{bsz}_m150_tokens"][i:i + bsz, 0]):
            same += upto_eos(a) == upto_eos(b); total += 1
print("greedy", same, "/", total, flush=True)
gold = load_npz("gen_beam.npz")
s2 = t2 = 0
for bsz, beam in ((1, 5), (4, 5), (10, 3), (5, 10)):
    g = BeamSearchOracle(model, beam, 150, PAD, BOS, EOS)
    for bi, i in enumerate(range(0, 10, bsz)):
        out = g.generate(src[i:i + bsz]).numpy()
        ref = gold[f"b{bsz}_k{beam}_batch{bi}"]
        for b in range(out.shape[0]):
            for k in range(out.shape[1]):
                s2 += upto_eos(out[b, k]) == upto_eos(ref[b, k]) if out.shape == ref.shape else 0; t2 += 1
print("beam", s2, "/", t2, flush=True)
gold = load_npz("gen_spec_beam.npz")
for smart in (False, True):
    s3 = t3 = 0
    ci = 0
    while f"smart{int(smart)}_case{ci}_rows" in gold:
        key = f"smart{int(smart)}_case{ci}"
        rows = gold[key + "_rows"].tolist()
        bsz, nbest, N, D = gold[key + "_params"].tolist()
        g = BeamSearchSpeculativeOracle(model, 150, nbest, D, N, V, smart, PAD, BOS, EOS, c, max_steps=400)
        for bi, i in enumerate(range(0, len(rows), bsz)):
            sel = src[rows[i:i + bsz]]
            out = g.generate(sel[:, :int((sel != PAD).sum(1).max())]).numpy()
            ref = gold[f"{key}_batch{bi}"]
            for b in range(out.shape[0]):
                for k in range(out.shape[1]):
                    s3 += upto_eos(out[b, k]) == upto_eos(ref[b, k]); t3 += 1
        ci += 1
    print("beam-speculative smart", smart, s3, "/", t3, flush=True)
