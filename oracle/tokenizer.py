"""ORACLE (test infrastructure): restatement of the reference SMILES tokenizer.

  split_smiles  <- SimpleSmilesTokenizer.split_into_tokens   src/data_handling/tokenizer_smiles.py:8-19
  encode        <- ChemSMILESTokenizer.encode                 src/data_handling/tokenizer_smiles.py:34-39
  decode        <- GenericTokenizer.decode                    src/data_handling/tokenizer_base.py:80-91

Pinned by tests/golden/tokenizer_cases.json (outputs of the reference tokenizer itself).  Also serves as the
CPU baseline of tests/bench_tokenizer.py (kind "port": the same `re` engine and per-token dict lookups as the
reference).
"""
from __future__ import annotations

import re

PAD, BOS, EOS, UNK = 0, 1, 2, 3

# alternatives in the reference's order: bracket atom | Br | Cl | one-letter atoms | bonds, branches, dots ... | %NN | digit
_PIECE = re.compile(r"(\[[^\]]+]|Br?|Cl?|N|O|S|P|F|I|b|c|n|o|s|p|\(|\)|\.|=|#|-|\+|\\|\/|:|~|@|\?|>|\*|\$|\%[0-9]{2}|[0-9])")


def split_smiles(smi: str) -> list[str]:
    return _PIECE.findall(smi)


def encode(vocab: dict[str, int], smi: str) -> list[int]:
    return [BOS] + [vocab.get(tok, UNK) for tok in split_smiles(smi)] + [EOS]


def decode(inverse_vocab: dict[int, str], ids) -> str:
    out = []
    for i in ids:
        i = int(i)
        if i not in (BOS, EOS, PAD):
            out.append(inverse_vocab[i])
        if i == EOS:
            break
    return "".join(out)
