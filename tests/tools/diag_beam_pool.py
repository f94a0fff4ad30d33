"""Diagnostic (GPU box): pooled vs per-batch beam-speculative outputs on the tiny model, hypothesis by hypothesis."""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import translation_transformer_amd as tta
from util_models import tiny_state, fixture_tokens, upto_eos, PAD, BOS, EOS

st, cfg = tiny_state()
native = tta.NativeTransformer(st, cfg["num_heads"], 0, device=0)
src, _, c, V = fixture_tokens()


def batches_of(groups):
    out = []
    for rows in groups:
        sel = src[rows]
        out.append(sel[:, :int((sel != PAD).sum(1).max())].cuda())
    return out


def run(groups, params, smart, cap):
    max_len, nbest, D, N = params
    bs = batches_of(groups)
    one = tta.TranslationInferenceBeamSearchSpeculative(native, max_len, nbest, D, N, V, smart, PAD, BOS, EOS, c, max_steps=300)
    ref = [one.generate(b) for b in bs]
    many = tta.TranslationInferenceBeamSearchSpeculative(native, max_len, nbest, D, N, V, smart, PAD, BOS, EOS, c, max_steps=300)
    out = many.generate_many(bs, in_flight=3, pool=True, capacity=cap)
    bad = 0
    for bi, (a, b) in enumerate(zip(out, ref)):
        if a.shape != b.shape:
            print("   shape", bi, tuple(a.shape), tuple(b.shape)); bad += 1; continue
        a, b = a.cpu().numpy(), b.cpu().numpy()
        for s in range(a.shape[0]):
            for k in range(a.shape[1]):
                if not (a[s, k] == b[s, k]).all():
                    bad += 1
                    if bad <= 6:
                        pos = int(np.argmax(a[s, k] != b[s, k]))
                        print(f"   batch {bi} (rows {groups[bi]}) source {s} rank {k}: first diff at {pos}: pool {a[s, k][max(0,pos-3):pos+4].tolist()} "
                              f"per-batch {b[s, k][max(0,pos-3):pos+4].tolist()}; same up to EOS: {upto_eos(a[s, k]) == upto_eos(b[s, k])}; "
                              f"pool row is a permutation of per-batch ranks: {any((a[s, k] == b[s, kk]).all() for kk in range(a.shape[1]))}")
    cnt = {n: (getattr(many, n), getattr(one, n)) for n in ("model_calls_num", "accepted_tokens_num", "produced_non_pad_tokens", "model_input_lines_num", "b_sz")}
    print(f"groups={groups} params={params} smart={smart} cap={cap}: differing hypotheses {bad}; counters (pool, per-batch) {cnt}; "
          f"as_given {many.stats_total.get('batches_decoded_as_given')} device iters {many.stats_total.get('device_model_calls')}", flush=True)


import os
P = (150, 5, 10, 3)
for groups in ([[5, 6]], [[5]], [[6]], [[5, 6], [8, 9, 0]], [[0, 2, 3, 4], [5, 6]], [[0, 2, 3, 4], [8, 9, 0]], [[4], [9]], [[4], [5]], [[0, 2, 3, 4], [9]]):
    for cap in (64, 2, 1):
        run(groups, P, False, cap)
os.environ["TTX_POOL_SESSIONS"] = "1"
print("one session:")
run([[0, 2, 3, 4], [5, 6], [8, 9, 0]], P, False, 2)
run([[0, 2, 3, 4], [5, 6], [8, 9, 0]], P, False, 64)
os.environ["TTX_NO_GRAPH"] = "1"
print("one session, no graphs (new model):")
native = tta.NativeTransformer(st, cfg["num_heads"], 0, device=0)
run([[0, 2, 3, 4], [5, 6], [8, 9, 0]], P, False, 2)
