"""The C++ SMILES tokenizer / collate / detokenizer of libttx_hip.so against the reference tokenizer's own outputs
(tests/golden/tokenizer_cases.json: fixture lines plus synthetic strings covering every regex alternative, unknown
characters, unterminated brackets, %NN ring closures)."""
import json

import numpy as np
import pytest

import translation_transformer_amd as tta
from util_models import GOLDEN


@pytest.fixture(scope="module")
def cases():
    return json.loads((GOLDEN / "tokenizer_cases.json").read_text())


@pytest.fixture(scope="module")
def tkz(cases):
    t = tta.NativeSmilesTokenizer()
    t.assign_vocab({v: int(k) for k, v in cases["vocab"].items()})
    return t


def test_encode_matches_reference(tkz, cases):
    assert tkz.n_tokens == len(cases["vocab"])
    for line, ids in zip(cases["lines"], cases["ids"]):
        assert tkz.encode(line) == ids, line


def test_decode_matches_reference(tkz, cases):
    for ids, dec, raw in zip(cases["ids"], cases["decoded"], cases["decoded_with_service"]):
        assert tkz.decode(np.array(ids)) == dec
        assert tkz.decode(ids, skip_service_tokens=False) == raw
    # decoding stops at the first EOS and ignores PAD/BOS, as PredictionWriter relies on (callbacks.py:55-64)
    assert tkz.decode([1, 4, 0, 4, 2, 4, 4]) == tkz.decoder_dict[4] * 2
    with pytest.raises(KeyError):
        tkz.decode([1, 9999, 2])


def test_encode_batch_is_pad_sequence_collate(tkz, cases):
    lines = cases["lines"][:10]
    got = tkz.encode_batch(lines)
    width = max(len(i) for i in cases["ids"][:10])
    assert got.shape == (10, width) and got.dtype == np.int64
    for row, ids in zip(got, cases["ids"][:10]):
        assert row[:len(ids)].tolist() == ids and (row[len(ids):] == 0).all()


def test_fixture_round_trip(tkz, cases):
    for line in cases["lines"][:20]:          # the 20 fixture SMILES are fully covered by the vocabulary
        assert tkz.decode(tkz.encode(line)) == line


def test_oracle_tokenizer_matches_reference(cases):
    from oracle import tokenizer as ot
    vocab = {v: int(k) for k, v in cases["vocab"].items()}
    inv = {int(k): v for k, v in cases["vocab"].items()}
    for line, pieces, ids, dec in zip(cases["lines"], cases["pieces"], cases["ids"], cases["decoded"]):
        assert ot.split_smiles(line) == pieces
        assert ot.encode(vocab, line) == ids
        assert ot.decode(inv, ids) == dec
