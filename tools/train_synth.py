"""Train synthetic-data weights for the benchmark and the full-size GPU tests.

Stock ``torch.nn.Transformer`` with the reference's hyper-parameters and state-dict key layout (SURVEY.md
§8(b) B6), the reference's loss (mean cross-entropy, no ignore_index: lightning_model.py:68) and Adam.
This is set-up tooling — it produces the weights the measured HIP path and the CPU oracle both load; it is
not part of the product path and nothing here is timed.
"""
from __future__ import annotations

import math
import sys
import time
from pathlib import Path

import numpy as np
import torch
from torch import nn

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tools.synth import SynthReactions, pad_batch, PAD, V  # noqa: E402


class _Emb(nn.Module):
    def __init__(self, vocab, d):
        super().__init__()
        self.embedding = nn.Embedding(vocab, d, padding_idx=PAD)


class TrainModel(nn.Module):
    def __init__(self, vocab=V, d=256, heads=8, ff=2048, n_enc=4, n_dec=4, dropout=0.0):
        super().__init__()
        self.src_token_featurizer = _Emb(vocab, d)
        self.tgt_token_featurizer = self.src_token_featurizer          # share_embeddings: true
        enc = nn.TransformerEncoder(nn.TransformerEncoderLayer(d, heads, ff, dropout, "relu", 1e-5, True, False), n_enc,
                                    nn.LayerNorm(d, eps=1e-5), enable_nested_tensor=False)
        dec = nn.TransformerDecoder(nn.TransformerDecoderLayer(d, heads, ff, dropout, "relu", 1e-5, True, False), n_dec,
                                    nn.LayerNorm(d, eps=1e-5))
        self.transformer = nn.Transformer(d_model=d, nhead=heads, batch_first=True, custom_encoder=enc, custom_decoder=dec)
        self.next_token_classifier = nn.Linear(d, vocab)
        pos = torch.arange(0, 5000, dtype=torch.float).unsqueeze(1)
        div = torch.exp(torch.arange(0, d, 2).float() * (-math.log(10000.0) / d))
        pe = torch.zeros(5001, d)
        pe[1:, 0::2] = torch.sin(pos * div)
        pe[1:, 1::2] = torch.cos(pos * div)
        self.register_buffer("pe", pe, persistent=False)

    def forward(self, src, tgt):
        s = self.src_token_featurizer.embedding(src) + self.pe[1:src.size(1) + 1]
        t = self.tgt_token_featurizer.embedding(tgt) + self.pe[1:tgt.size(1) + 1]
        causal = self.transformer.generate_square_subsequent_mask(tgt.size(1)).to(s.device)
        h = self.transformer(s, t, tgt_mask=causal, src_key_padding_mask=src == PAD, tgt_key_padding_mask=tgt == PAD,
                             memory_key_padding_mask=src == PAD)
        return self.next_token_classifier(h)


def train(kind="mit", steps=1500, batch=128, lr=1e-3, warmup=150, seed=123456, n_enc=4, n_dec=4, device="cuda",
          log_every=100, n_train=50000, verbose=True) -> dict:
    torch.manual_seed(seed)
    data = SynthReactions(seed + 1, kind)
    src, tgt = data.dataset(n_train)
    model = TrainModel(n_enc=n_enc, n_dec=n_dec).to(device)
    opt = torch.optim.Adam(model.parameters(), lr=lr, betas=(0.9, 0.98))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: min((s + 1) / warmup, math.sqrt(warmup / (s + 1))))
    crit = nn.CrossEntropyLoss(reduction="mean")
    rng = np.random.default_rng(seed)
    model.train()
    t0 = time.time()
    for step in range(steps):
        idx = rng.integers(0, n_train, size=batch)
        s = torch.from_numpy(pad_batch([src[i] for i in idx])).to(device)
        t = torch.from_numpy(pad_batch([tgt[i] for i in idx])).to(device)
        logits = model(s, t[:, :-1])
        loss = crit(logits.reshape(-1, V), t[:, 1:].reshape(-1))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        sched.step()
        if verbose and (step % log_every == 0 or step == steps - 1):
            print(f"[train_synth] step {step} loss {loss.item():.4f} ({time.time() - t0:.1f}s)", flush=True)
    model.eval()
    return {k: v.detach().float().cpu() for k, v in model.state_dict().items()}


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="mit")
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--out", default="gpurun_out/synth_weights.pt")
    a = ap.parse_args()
    sd = train(a.kind, a.steps, a.batch, a.lr)
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    torch.save(sd, a.out)
    print("saved", a.out)
