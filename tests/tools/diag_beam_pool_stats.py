"""Diagnostic (GPU box): what the beam source pool sees on the bench workloads c3 / c4 — per-source status, iterations,
longest hypothesis, how many batches are sent back to be decoded as given — and timings of pool vs as-given."""
import os, sys, time
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import translation_transformer_amd as tta
import bench
from tools.synth import SynthReactions, batches, PAD, BOS, EOS, C_TOK, V

cfgname = sys.argv[1] if len(sys.argv) > 1 else "c4"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 32
bc = bench.BEAM_CONFIGS[cfgname]
sd = bench.get_weights(1500, "cuda:0", bc["kind"], bc["layers"], {})
model = tta.NativeTransformer(sd, num_heads=8, pad_token_idx=PAD, device=0)
src_all, _ = SynthReactions(123456, bc["kind"]).dataset((nb + 2) * bc["bs"])
bs = [torch.from_numpy(b).cuda() for b in batches(src_all, bc["bs"])]
warm, timed = bs[:2], bs[2:]
K, N, D, L = bc["n_best"], bc["N"], 10, 200
for smart in (False, True):
    mk = lambda: tta.TranslationInferenceBeamSearchSpeculative(model, L, K, D, N, V, smart, PAD, BOS, EOS, C_TOK, max_steps=4 * L)
    g = mk(); g.generate_many(warm * 4, in_flight=8)
    g = mk(); g.generate_many(warm * 4, in_flight=8, pool=False)
    for pool in (True, False):
        g = mk()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        outs = g.generate_many(timed, in_flight=8, pool=pool)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        lens = torch.cat([(o != PAD).sum(-1).flatten() for o in outs]).float()
        eos = torch.cat([(o == EOS).any(-1).flatten() for o in outs]).float()
        print(f"{cfgname} smart={smart} pool={pool}: {len(timed) * bc['bs'] / dt:7.1f} reactions/s, {dt*1e3:.0f} ms; calls {g.model_calls_num}; "
              f"device iterations {g.stats_total.get('device_model_calls')}; "
              f"hypothesis length mean {lens.mean():.1f} max {lens.max():.0f}; rows with EOS {eos.mean():.3f}; out widths {sorted(set(int(o.shape[2]) for o in outs))[-3:]}", flush=True)
