"""The oracle's model restatement against outputs of the reference itself (tests/golden/*_model_io.npz)."""
import numpy as np
import torch

from oracle.model import OracleTransformer, config_from_state
from util_models import load_npz, tiny_state, full_state, state_shapes

TOL = 5e-5  # absolute, logits are O(10); the reference's own fast/slow paths differ by ~1e-6


def test_tiny_encode_decode_match_reference():
    st, cfg = tiny_state()
    m = OracleTransformer(config_from_state(st, cfg["num_heads"]), st)
    io = load_npz("tiny_model_io.npz")
    src = torch.from_numpy(io["src"])
    mask = src == 0
    mem = m.encode_src(src, mask)
    ref_mem = torch.from_numpy(io["memory"])
    assert (mem - ref_mem)[~mask].abs().max() < TOL
    assert float(mem[mask].abs().max()) == 0.0
    lg = m.decode_tgt(torch.from_numpy(io["tgt_in"]), ref_mem, mask)
    assert (lg - torch.from_numpy(io["logits"])).abs().max() < TOL
    lg = m.decode_tgt(torch.from_numpy(io["tgt_ragged"]), ref_mem, mask)
    assert (lg - torch.from_numpy(io["logits_ragged"])).abs().max() < TOL
    fwd = m(src, torch.from_numpy(io["tgt_in"][:, :1]))
    assert (fwd - torch.from_numpy(io["fwd_bos"])).abs().max() < TOL


def test_full_size_seeded_weights_match_reference():
    io = load_npz("full_model_io.npz")
    V = int(io["vocab_size"])
    st = full_state(V, int(io["seed"]))
    assert [n for n, _ in state_shapes(V, 256, 2048, 4, 4)] == [str(x) for x in io["weight_names"]]
    assert abs(sum(float(np.abs(v).sum()) for v in st.values()) - float(io["weight_checksum"])) < 1e-3
    m = OracleTransformer(config_from_state(st, 8), st)
    src, tgt = torch.from_numpy(io["src"]), torch.from_numpy(io["tgt"])
    mask = src == 0
    mem = m.encode_src(src, mask)
    ref_mem = torch.from_numpy(io["memory"])
    assert (mem - ref_mem)[~mask].abs().max() < TOL
    lg = m.decode_tgt(tgt, ref_mem, mask)
    ref = torch.from_numpy(io["logits"])
    assert (lg - ref).abs().max() < 2e-4 * max(1.0, float(ref.abs().max()))
    assert torch.equal(lg.argmax(-1), ref.argmax(-1))
