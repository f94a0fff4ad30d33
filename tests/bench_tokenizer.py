#!/usr/bin/env python3
"""Throughput of the C++ tokenizer / collate / detokenizer against the CPU port of the reference tokenizer
(oracle/tokenizer.py, same `re` engine as the reference) on the fixture reactions repeated.  Host-only."""
from __future__ import annotations

import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402


def main():
    import translation_transformer_amd as tta
    from oracle import tokenizer as ot
    cases = json.loads((ROOT / "tests/golden/tokenizer_cases.json").read_text())
    vocab = {v: int(k) for k, v in cases["vocab"].items()}
    inv = {int(k): v for k, v in cases["vocab"].items()}
    lines = cases["lines"][:20] * 2000                       # 40 000 SMILES, the size of the USPTO-MIT test set
    tkz = tta.NativeSmilesTokenizer()
    tkz.assign_vocab(vocab)
    t0 = time.perf_counter()
    ref_ids = [ot.encode(vocab, l) for l in lines]
    t_ref_enc = time.perf_counter() - t0
    t0 = time.perf_counter()
    batches = [tkz.encode_batch(lines[i:i + 32]) for i in range(0, len(lines), 32)]
    t_nat_enc = time.perf_counter() - t0
    for i in range(0, 64, 32):
        for row, ids in zip(batches[i // 32], ref_ids[i:i + 32]):
            assert row[:len(ids)].tolist() == ids
    flat = [np.array(i, dtype=np.int64) for i in ref_ids]
    t0 = time.perf_counter()
    ref_dec = [ot.decode(inv, i) for i in flat]
    t_ref_dec = time.perf_counter() - t0
    t0 = time.perf_counter()
    nat_dec = tkz.decode_batch(flat)
    t_nat_dec = time.perf_counter() - t0
    assert nat_dec == ref_dec
    n = len(lines)
    print(json.dumps({"lines": n, "encode_lines_per_s": {"cpp_batched": n / t_nat_enc, "python_port": n / t_ref_enc},
                      "decode_lines_per_s": {"cpp": n / t_nat_dec, "python_port": n / t_ref_dec}}))


if __name__ == "__main__":
    main()
