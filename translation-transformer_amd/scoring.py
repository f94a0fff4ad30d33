"""Top-N accuracy of a prediction CSV — what src/score_predictions.py:15-57 reports, without
requiring RDKit at run time.

The CSV is what the PredictionWriter appends (src/callbacks.py:55-64): ``source,target,prediction_1..N`` per line,
split on commas exactly like the reference does (:20-24; a header line, if present, is scored as a row by the
reference too — pass ``skip_header=True`` to drop it).  A prediction counts as a hit when it equals the target
after canonicalisation.  With RDKit importable the canonical form is RDKit's (invalid SMILES -> "!", :7-13), which is
the reference's definition; without it strings are compared as they are (exact-string top-N: a lower bound of the
RDKit figure, since two spellings of one molecule do not match) and the invalid-SMILES rate is reported as None.
`score_tokens` does the same on token ids, on the device, straight from a generator's output.

Parity: RDKit is absent from the build image and the reference's own script cannot run there — the RDKit branch
is "parity unpinned"; the exact-string branch is checked against a plain per-row restatement in
tests/test_scoring.py.
"""
from __future__ import annotations

import argparse

TOP_N = (1, 3, 5, 10, 15, 20, 50)


def _rdkit_canonicalizer():
    try:
        from rdkit import Chem, RDLogger
    except Exception:
        return None
    RDLogger.DisableLog("rdApp.*")

    def canon(s: str) -> str:
        if s == "":
            return s
        m = Chem.MolFromSmiles(s)
        return "!" if m is None else Chem.MolToSmiles(m)
    return canon


def score_rows(targets: list, predictions: list, canonicalize=None) -> dict:
    """targets: list[str]; predictions: list[list[str]] (ragged; missing ranks count as empty strings, :25)."""
    n = len(targets)
    width = max((len(p) for p in predictions), default=0)
    canon = canonicalize or (lambda s: s)
    cache: dict = {}

    def c(s):
        v = cache.get(s)
        if v is None:
            v = cache[s] = canon(s)
        return v

    first_hit = []                                   # rank (1-based) of the first matching prediction, 0 = none
    invalid = [0] * width
    empty = [0] * width
    for t, ps in zip(targets, predictions):
        ct = c(t)
        rank = 0
        for i in range(width):
            cp = c(ps[i]) if i < len(ps) else ""
            if cp == "!":
                invalid[i] += 1
            if cp == "":
                empty[i] += 1
            if rank == 0 and cp == ct:
                rank = i + 1
        first_hit.append(rank)
    ks = [k for k in TOP_N if k <= width]
    acc = {f"top {k}": 100.0 * sum(1 for r in first_hit if 0 < r <= k) / max(n, 1) for k in ks}
    inv = {f"prediction {k}": (100.0 * invalid[k - 1] / max(n, 1)) if canonicalize else None for k in ks}
    emp = {f"prediction {k}": 100.0 * empty[k - 1] / max(n, 1) for k in ks}
    return {"n_queries": n, "n_preds": width, "accuracy": acc, "invalid_smiles": inv, "empty_smiles": emp,
            "canonicalizer": "rdkit" if canonicalize else "exact string"}


def score_csv(filename: str, skip_header: bool = False, canonicalize="auto") -> dict:
    with open(filename) as f:
        lines = [ln.strip() for ln in f.readlines()]
    if skip_header and lines and lines[0].startswith("source,target"):
        lines = lines[1:]
    targets, preds = [], []
    for line in lines:
        _, t, *ps = line.split(",")
        targets.append(t)
        preds.append(ps)
    canon = _rdkit_canonicalizer() if canonicalize == "auto" else canonicalize
    return score_rows(targets, preds, canon)


def score_tokens(pred, tgt, pad: int, bos: int, eos: int) -> dict:
    """Exact token-id top-N on tensors: pred Long[B,N,L] (hypotheses best-first), tgt Long[B,Lt].  A hypothesis is
    compared on its tokens up to the first EOS with BOS/PAD skipped, like GenericTokenizer.decode
    (src/data_handling/tokenizer_base.py:80-91).  Runs on whatever device the tensors are on."""
    import torch

    def clean(x):                                    # [..., L] -> same shape, tokens kept in order, rest = -1
        after = (x == eos).cumsum(-1) > 0            # from the first EOS on (EOS itself is dropped too)
        keep = ~after & (x != bos) & (x != pad)
        # stable compaction: kept tokens first, in their original order
        order = torch.argsort((~keep).to(torch.int8), dim=-1, stable=True)
        return torch.where(keep.gather(-1, order), x.gather(-1, order), torch.full_like(x, -1))

    B, N, L = pred.shape
    W = max(L, tgt.shape[1])
    p = torch.full((B, N, W), pad, dtype=pred.dtype, device=pred.device)
    p[:, :, :L] = pred
    t = torch.full((B, W), pad, dtype=pred.dtype, device=pred.device)
    t[:, :tgt.shape[1]] = tgt.to(pred.device)
    hit = (clean(p) == clean(t).unsqueeze(1)).all(-1)                  # [B, N]
    top = hit.cumsum(1) > 0
    ks = [k for k in TOP_N if k <= N]
    return {"n_queries": B, "n_preds": N, "accuracy": {f"top {k}": 100.0 * float(top[:, k - 1].float().mean()) for k in ks}}


def _fmt(d: dict) -> str:
    w = max(len(k) for k in d)
    return "\n".join(f"{k.ljust(w)}    {'n/a' if v is None else round(v, 6)}" for k, v in d.items())


def main(argv=None):
    ap = argparse.ArgumentParser(description="top-N accuracy of a prediction CSV")
    ap.add_argument("--filename", "-f", type=str, required=True)
    ap.add_argument("--skip-header", action="store_true")
    a = ap.parse_args(argv)
    r = score_csv(a.filename, skip_header=a.skip_header)
    print(f"({r['n_queries']} queries, {r['n_preds']} predictions each, match on {r['canonicalizer']})")
    print("Accuracy, %")
    print(_fmt(r["accuracy"]))
    print()
    print("Invalid SMILES, %")
    print(_fmt(r["invalid_smiles"]))
    print()
    print("Empty SMILES, %")
    print(_fmt(r["empty_smiles"]))
    return r


if __name__ == "__main__":
    main()
