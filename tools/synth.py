"""Synthetic USPTO-shaped reaction data: there is no network, hence no USPTO files, no
vocabulary and no checkpoint (the reference downloads them from Google Drive, README.md:34-79).

Vocabulary (V = 256): 0 PAD, 1 BOS, 2 EOS, 3 UNK as in the reference tokenizer (tokenizer_base.py:27-30);
4 = "c", the most frequent regular token, used as the draft replacement token (lightning_model.py:117);
5 = ".", the fragment separator; 6 = a "branch" token that rewrites the token after it; 7 = a "bond" token
inserted where two kept fragments are joined; 8..255 regular tokens with Zipf(1.2) frequencies.

A source is 2-5 "."-separated fragments; some are reagents/solvents drawn from a small fixed library
(they never appear in the product, like the Cl / CCOCC / [Na+] of the fixture reactions), the others are
reactants.  The target ("product") is the reactant fragments in source order joined by the bond token,
with the token after every branch token replaced through a fixed permutation.  So most of the target is a
verbatim copy of source spans — which is what makes the reference's copy-drafts work — interrupted at
deterministic, learnable edit sites.  Lengths follow the fixture statistics: source length (with BOS/EOS)
~ clip(LogNormal(ln 62, 0.42), 20, 180) for the MIT-shaped set, shorter sources and longer targets for the
50K-shaped (retrosynthesis) set.
"""
from __future__ import annotations

import numpy as np

PAD, BOS, EOS, UNK, C_TOK, DOT, BRANCH, BOND = 0, 1, 2, 3, 4, 5, 6, 7
V = 256
FIRST_REGULAR = 8


def _zipf_probs(n: int, s: float = 1.2) -> np.ndarray:
    p = 1.0 / np.arange(1, n + 1) ** s
    return p / p.sum()


class SynthReactions:
    def __init__(self, seed: int = 123456, kind: str = "mit"):
        assert kind in ("mit", "50k")
        self.kind = kind
        self.rng = np.random.default_rng(seed)
        lib_rng = np.random.default_rng(987654321)          # the library and permutation are fixed
        self.p_tok = _zipf_probs(V - FIRST_REGULAR)
        self.library = [self._draw(lib_rng, int(lib_rng.integers(2, 13)), allow_branch=False) for _ in range(40)]
        perm = lib_rng.permutation(V - FIRST_REGULAR)
        self.branch_map = np.arange(V)
        self.branch_map[FIRST_REGULAR:] = FIRST_REGULAR + perm
        self.branch_map[C_TOK] = FIRST_REGULAR + int(perm[0])

    def _draw(self, rng, n: int, allow_branch: bool = True) -> list[int]:
        toks = FIRST_REGULAR + rng.choice(V - FIRST_REGULAR, size=n, p=self.p_tok)
        toks = toks.tolist()
        out = []
        for t in toks:
            u = rng.random()
            if u < 0.18:
                out.append(C_TOK)
            elif allow_branch and u < 0.24 and out and out[-1] != BRANCH:
                out.append(BRANCH)
            else:
                out.append(int(t))
        if out and out[-1] == BRANCH:
            out[-1] = C_TOK
        return out

    def _length(self) -> int:
        if self.kind == "mit":
            L = self.rng.lognormal(np.log(62.0), 0.42)
            return int(np.clip(round(L), 20, 180))
        L = self.rng.lognormal(np.log(52.0), 0.38)
        return int(np.clip(round(L), 14, 120))

    def pair(self) -> tuple[list[int], list[int]]:
        rng = self.rng
        body = self._length() - 2
        n_react = int(rng.integers(1, 4)) if self.kind == "mit" else int(rng.integers(1, 3))
        n_reag = int(rng.integers(1, 5)) if self.kind == "mit" else 1
        reag = [self.library[int(rng.integers(0, len(self.library)))] for _ in range(n_reag)]
        left = body - sum(len(r) for r in reag) - (n_react + n_reag - 1)
        left = max(left, 3 * n_react)
        cuts = np.sort(rng.choice(np.arange(1, left), size=n_react - 1, replace=False)) if n_react > 1 else np.array([], int)
        sizes = np.diff(np.concatenate([[0], cuts, [left]])).astype(int)
        react = [self._draw(rng, int(s)) for s in sizes]
        if self.kind == "50k":
            # retrosynthesis: the "leaving group" written next to the reactants is a fixed function of the
            # first reactant token, so it is predictable from the product
            reag = [self.library[react[0][0] % len(self.library)]]
        frags = [("r", f) for f in react] + [("g", f) for f in reag]
        order = rng.permutation(len(frags))
        src = [BOS]
        kept = []
        for oi, i in enumerate(order):
            kind, f = frags[int(i)]
            if oi:
                src.append(DOT)
            src.extend(f)
            if kind == "r":
                kept.append(f)
        src.append(EOS)
        tgt = [BOS]
        for fi, f in enumerate(kept):
            if fi:
                tgt.append(BOND)
            prev = None
            for t in f:
                tgt.append(int(self.branch_map[t]) if prev == BRANCH else t)
                prev = t
        tgt.append(EOS)
        if self.kind == "50k":          # retrosynthesis direction: product -> reactants
            src, tgt = tgt, src
        return src, tgt

    def dataset(self, n: int) -> tuple[list[list[int]], list[list[int]]]:
        pairs = [self.pair() for _ in range(n)]
        return [p[0] for p in pairs], [p[1] for p in pairs]


def pad_batch(rows: list[list[int]], pad: int = PAD) -> np.ndarray:
    L = max(len(r) for r in rows)
    out = np.full((len(rows), L), pad, dtype=np.int64)
    for i, r in enumerate(rows):
        out[i, :len(r)] = r
    return out


def batches(rows: list[list[int]], batch_size: int):
    """Unshuffled fixed-size batches, each padded to its own longest row — what the reference's predict
    dataloader + pad_sequence collate produce (seq2seq_wrappers.py:121-127,168-175)."""
    for i in range(0, len(rows), batch_size):
        yield pad_batch(rows[i:i + batch_size])


def smiles_vocabulary() -> dict:
    """A token string for every id of the synthetic vocabulary, so that the string side of the pipeline (regex tokenizer,
    collate, detokenizer, CSV writer: src/data_handling/tokenizer_smiles.py:8,34-39, tokenizer_base.py:80-94,
    src/callbacks.py:49-64) can be driven with the same reactions: {token string: id} like the reference's vocab.json read
    back (tokenizer_base.py:60-66).  Every string is ONE token of the reference's SMILES regex whatever its neighbours are
    (single-character atoms / bonds / ring digits, two-digit ring closures %NN, bracket atoms), so decode(ids) tokenizes
    back to the same ids."""
    singles = list("NOSPFIbcnosp()=#-+\\/:~@>*$") + [str(d) for d in range(10)]      # '.', 'c' placed below; '?' is <UNK>
    toks = {"<PAD>": PAD, "<BOS>": BOS, "<EOS>": EOS, "?": UNK, "c": C_TOK, ".": DOT, "(": BRANCH, "=": BOND}
    pool = [t for t in singles if t not in toks] + ["Cl", "Br", "C", "B"] + [f"%{i:02d}" for i in range(100)]
    elems = ["C", "N", "O", "S", "P", "Si", "B", "Se", "Na", "K", "Li", "Mg", "Zn", "Cu", "Pd", "Sn"]
    decor = ["@H", "@@H", "H", "+", "-", "H2", "H3", "H+", "2+", ":1", ":2", ":3"]
    pool += [f"[{e}{d}]" for d in decor for e in elems]
    ids = [i for i in range(FIRST_REGULAR, V)]
    assert len(pool) >= len(ids) and len(set(pool)) == len(pool)
    for i, t in zip(ids, pool):
        toks[t] = i
    assert len(toks) == V
    return toks
