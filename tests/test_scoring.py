"""translation_transformer_amd.scoring against a plain per-row restatement of src/score_predictions.py:15-57
(exact-string branch; RDKit is absent here, see the module docstring) and on the fixture targets."""
import random

import numpy as np
import torch

import translation_transformer_amd  # noqa: F401
from translation_transformer_amd import scoring
from util_models import fixture_tokens, PAD, BOS, EOS


def restated(targets, preds):
    """The reference's DataFrame algebra row by row: hit_i, cumulative or, column means."""
    width = max(len(p) for p in preds)
    rows = [[p[i] if i < len(p) else "" for i in range(width)] for p in preds]
    hit_top = np.zeros((len(targets), width), dtype=bool)
    for r, (t, ps) in enumerate(zip(targets, rows)):
        seen = False
        for i, p in enumerate(ps):
            seen = seen or (p == t)
            hit_top[r, i] = seen
    ks = [k for k in (1, 3, 5, 10, 15, 20, 50) if k <= width]
    acc = {f"top {k}": 100.0 * hit_top[:, k - 1].mean() for k in ks}
    emp = {f"prediction {k}": 100.0 * np.mean([ps[k - 1] == "" for ps in rows]) for k in ks}
    return acc, emp


def test_csv_scoring_matches_restatement(tmp_path):
    rng = random.Random(7)
    alphabet = ["C", "c1ccccc1", "CC(=O)O", "N", "O=C", "CCO", "Br", "Cl"]
    lines, targets, preds = [], [], []
    for _ in range(200):
        t = rng.choice(alphabet)
        n = rng.choice([1, 3, 5, 12])
        ps = [rng.choice(alphabet + [""]) for _ in range(n)]
        targets.append(t)
        preds.append(ps)
        lines.append(",".join(["src", t] + ps))
    f = tmp_path / "pred.csv"
    f.write_text("\n".join(lines) + "\n")
    got = scoring.score_csv(str(f), canonicalize=None)
    acc, emp = restated(targets, preds)
    assert got["n_queries"] == 200 and got["n_preds"] == 12
    assert list(got["accuracy"]) == ["top 1", "top 3", "top 5", "top 10"]
    for k in acc:
        assert abs(got["accuracy"][k] - acc[k]) < 1e-9
    for k in emp:
        assert abs(got["empty_smiles"][k] - emp[k]) < 1e-9
    assert all(v is None for v in got["invalid_smiles"].values())       # no RDKit: not measurable
    scoring.main(["-f", str(f)])


def test_token_scoring():
    src, tgt, _, _ = fixture_tokens()
    B, Lt = tgt.shape
    pred = torch.full((B, 3, Lt + 5), PAD, dtype=torch.int64)
    pred[:, 1, :Lt] = tgt                        # rank 2 is right for every row ...
    pred[:4, 0, :Lt] = tgt[:4]                   # ... rank 1 for the first four
    pred[:, 2, 0] = BOS
    pred[:, 2, 1] = EOS                          # an empty hypothesis
    pred[4:, 0, 0] = BOS
    pred[4:, 0, 1] = 5
    pred[4:, 0, 2] = EOS
    r = scoring.score_tokens(pred, tgt, PAD, BOS, EOS)
    assert abs(r["accuracy"]["top 1"] - 100.0 * 4 / B) < 1e-4 and abs(r["accuracy"]["top 3"] - 100.0) < 1e-4
    # tokens after the first EOS are ignored, PAD/BOS are skipped
    junk = pred.clone()
    for b in range(B):
        n = int((tgt[b] != PAD).sum())
        junk[b, 1, n:n + 3] = 7
    assert scoring.score_tokens(junk, tgt, PAD, BOS, EOS)["accuracy"]["top 3"] == 100.0
