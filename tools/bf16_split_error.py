"""Feasibility of split-precision MFMA for the FFN pair: error of a decoder FFN layer of the trained
synthetic model computed from bf16 pieces (x = hi + lo (+ lo2), products of bf16 are exact in fp32, fp32 accumulation)
against float64.  CPU only.  Usage: python tools/bf16_split_error.py [weights.pt]"""
import sys
import torch, numpy as np
torch.manual_seed(0)
import os
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), '.weights_cache', 'synth_mit_1500.pt')
sd = torch.load(path, weights_only=True, map_location='cpu')
W1 = sd['transformer.decoder.layers.1.linear1.weight'].double(); b1 = sd['transformer.decoder.layers.1.linear1.bias'].double()
W2 = sd['transformer.decoder.layers.1.linear2.weight'].double(); b2 = sd['transformer.decoder.layers.1.linear2.bias'].double()
# LayerNorm-like inputs: unit-variance rows scaled by the layer's norm weights
g = sd['transformer.decoder.layers.1.norm2.weight'].double(); bb = sd['transformer.decoder.layers.1.norm2.bias'].double()
x = torch.randn(2048, 256, dtype=torch.float64) * g + bb
def split(t, parts):
    out=[]; r=t.float()
    for _ in range(parts):
        h=r.bfloat16().float(); out.append(h); r=r-h
    return out
def mm_split(a, b, parts, terms):
    A=split(a, parts); B=split(b, parts)
    acc=torch.zeros(a.shape[0], b.shape[0], dtype=torch.float32)
    for (i,j) in terms:
        acc += (A[i] @ B[j].T)          # fp32 accumulate of exact bf16 products (products of bf16 are exact in fp32)
    return acc
ref_h = torch.relu(x @ W1.T + b1)
ref_y = ref_h @ W2.T + b2
f32_h = torch.relu(x.float() @ W1.float().T + b1.float()); f32_y = f32_h @ W2.float().T + b2.float()
print("fp32      : h err %.2e  y err %.2e  (|y| max %.2f)" % ((f32_h.double()-ref_h).abs().max(), (f32_y.double()-ref_y).abs().max(), ref_y.abs().max()))
for name, parts, terms in (("bf16x1",1,[(0,0)]),("bf16x3",2,[(0,0),(0,1),(1,0)]),("bf16x4",2,[(0,0),(0,1),(1,0),(1,1)]),("bf16x6",3,[(0,0),(0,1),(1,0),(1,1),(0,2),(2,0)])):
    h = torch.relu(mm_split(x, W1, parts, terms) + b1.float())
    y = mm_split(h.double(), W2, parts, terms) + b2.float()
    print("%-9s : h err %.2e  y err %.2e" % (name, (h.double()-ref_h).abs().max(), (y.double()-ref_y).abs().max()))
