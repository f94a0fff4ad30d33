#!/usr/bin/env python3
"""Throughput of the beam paths (BASELINE.json configs[2] and configs[3]) on synthetic data — parity-test
configurations, not the bench.py headline.  Usage: python tools/bench_beam.py [--config c3|c4] [--batches K]"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3", choices=["c3", "c4"])
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--smart", type=int, default=0)
    ap.add_argument("--train-steps", type=int, default=int(os.environ.get("TTX_TRAIN_STEPS", "1500")))
    a = ap.parse_args()
    import translation_transformer_amd as tta
    from tools.synth import SynthReactions, batches, PAD, BOS, EOS, C_TOK, V
    from tools.train_synth import train
    kind, bs, nbest, N, D, n_enc = ("mit", 4, 5, 7, 10, 4) if a.config == "c3" else ("50k", 8, 10, 2, 10, 6)
    path = os.environ.get("TTX_WEIGHTS") if a.config == "c3" else os.environ.get("TTX_WEIGHTS_50K")
    path = path or f"/tmp/ttx_synth_{kind}_{a.train_steps}.pt"
    if os.path.exists(path):
        sd = torch.load(path, weights_only=True, map_location="cpu")
    else:
        sd = train(kind, steps=a.train_steps, n_enc=n_enc, n_dec=n_enc, device="cuda", verbose=False)
        torch.save(sd, path)
    model = tta.NativeTransformer(sd, 8, PAD, device=0)
    src, _ = SynthReactions(123456, kind).dataset((a.batches + 1) * bs)
    bt = [torch.from_numpy(b).cuda() for b in batches(src, bs)]
    gen = tta.TranslationInferenceBeamSearchSpeculative(model, 200, nbest, D, N, V, bool(a.smart), PAD, BOS, EOS, C_TOK,
                                                         max_steps=400)
    gen.generate(bt[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = [gen.generate(b) for b in bt[1:]]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = a.batches * bs
    print(json.dumps({"config": a.config, "algorithm": "beam_search_speculative", "smart_drafts_mode": bool(a.smart),
                      "batch_size": bs, "n_best": nbest, "n_drafts": N, "draft_len": D, "reactions": n,
                      "reactions_per_s": n / dt, "model_calls": gen.model_calls_num,
                      "acceptance_rate": gen.accepted_tokens_num / max(1, gen.produced_non_pad_tokens),
                      "rows_with_eos_top1": int(sum(int((o[:, 0] == EOS).any(1).sum()) for o in outs))}))


if __name__ == "__main__":
    main()
