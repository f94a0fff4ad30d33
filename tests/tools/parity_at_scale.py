#!/usr/bin/env python3
"""How often does a near-tie flip a token?  K batches of the bench workload (BASELINE config #2, trained synthetic weights)
through the CPU oracle (full-prefix recompute, torch fp32 — the reference's arithmetic) and through the HIP path: rows whose
tokens differ, and for each such row the gap between the two best logits of the ORACLE at the first differing position (a
flip is a near-tie if that gap is of the size of the fp32 noise between two implementations, ~1e-5).  Test infrastructure
(imports oracle/); not part of bench.py's default run: about 6 reactions/s on 16 host cores.
Usage: python tests/tools/parity_at_scale.py [--batches 24]"""
import argparse
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import translation_transformer_amd as tta  # noqa: E402
from bench import get_weights, usable_cores  # noqa: E402
from tools.synth import SynthReactions, batches, PAD, BOS, EOS, C_TOK  # noqa: E402
from oracle.model import OracleTransformer, config_from_state  # noqa: E402
from oracle.decoding import GreedySpeculativeOracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=24)
    ap.add_argument("--skip", type=int, default=5, help="leading batches left out (bench.py's warm-up batches)")
    a = ap.parse_args()
    sd = get_weights(1500, "cuda:0")
    src_all, _ = SynthReactions(123456, "mit").dataset((a.batches + a.skip) * 32)
    bs = [torch.from_numpy(b) for b in batches(src_all, 32)][a.skip:]
    torch.set_num_threads(usable_cores())
    om = OracleTransformer(config_from_state(sd, 8, PAD), sd)
    t0 = time.perf_counter()
    with torch.inference_mode():
        ref = []
        for k, b in enumerate(bs):
            try:
                ref.append(GreedySpeculativeOracle(om, 200, 10, 3, PAD, BOS, EOS, C_TOK).generate(b))
            except Exception:
                ref.append(None)
            if (k + 1) % 8 == 0:
                print(f"   oracle: {k + 1} / {len(bs)} batches, {time.perf_counter() - t0:.0f} s", flush=True)
    print(f"oracle: {sum(b.shape[0] for b in bs)} reactions in {time.perf_counter() - t0:.0f} s on {usable_cores()} threads", flush=True)
    m = tta.NativeTransformer(sd, 8, PAD, device=0)
    g = tta.TranslationInferenceGreedySpeculative(m, 200, 10, 3, PAD, BOS, EOS, C_TOK)
    res = [o.cpu() if o is not None else None for o in g.generate_many([b.cuda() for b in bs], in_flight=8, reorder=True, on_error="skip")]
    m.close()
    diff, total = [], 0
    for bi, (r, o) in enumerate(zip(ref, res)):
        if r is None or o is None:
            assert r is None and o is None, "one side raises, the other does not"
            continue
        total += r.shape[0]
        bad = (r[:, 0] != o[:, 0]).any(dim=1).nonzero().flatten().tolist()
        diff += [(bi, i) for i in bad]
    print(f"HIP path (fp32 MFMA, canonical slice order): {len(diff)} of {total} rows differ from the oracle", flush=True)
    for bi, i in diff[:12]:
        r, o = ref[bi][i, 0], res[bi][i, 0]
        pos = int((r != o).nonzero()[0])
        s = bs[bi][i:i + 1]
        with torch.inference_mode():
            mem = om.encode_src(s, s == PAD)
            lg = om.decode_tgt(r[None, :pos], mem, s == PAD)[0, -1]
        top = lg.topk(2)
        print(f"   batch {bi} row {i}: first difference at position {pos}: oracle token {int(r[pos])} vs {int(o[pos])}; oracle's two best "
              f"logits there {top.values[0]:.6f} ({int(top.indices[0])}) and {top.values[1]:.6f} ({int(top.indices[1])}): gap "
              f"{float(top.values[0] - top.values[1]):.2e}", flush=True)


if __name__ == "__main__":
    main()
