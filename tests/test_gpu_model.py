"""HIP model forward (through the C ABI) against the reference's own outputs (golden) and the oracle."""
import numpy as np
import pytest
import torch

from util_models import load_npz, tiny_state, full_state

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-3   # absolute; achieved values are printed


@pytest.fixture(scope="module")
def tta():
    import translation_transformer_amd as t
    assert t.lib().ttx_device_count() >= 1, "no gfx950 device: the HIP path must not be skipped silently"
    return t


@pytest.fixture(scope="module")
def tiny(tta):
    st, cfg = tiny_state()
    return tta.NativeTransformer(st, cfg["num_heads"], 0, device=0)


@pytest.fixture(scope="module")
def full(tta):
    io = load_npz("full_model_io.npz")
    return tta.NativeTransformer(full_state(int(io["vocab_size"]), int(io["seed"])), 8, 0, device=0)


def test_tiny_encode_matches_reference(tiny):
    io = load_npz("tiny_model_io.npz")
    src = torch.from_numpy(io["src"]).cuda()
    mem = tiny.encode_src(src, src == 0).cpu()
    ref = torch.from_numpy(io["memory"])
    mask = torch.from_numpy(io["src"]) == 0
    diff = (mem - ref)[~mask].abs().max().item()
    print("tiny memory max abs diff", diff)
    assert diff < 1e-4
    assert float(mem[mask].abs().max()) == 0.0


def test_tiny_decode_matches_reference(tiny):
    io = load_npz("tiny_model_io.npz")
    src = torch.from_numpy(io["src"])
    mem = torch.from_numpy(io["memory"]).cuda()
    mask = (src == 0).cuda()
    for tgt_key, out_key in (("tgt_in", "logits"), ("tgt_ragged", "logits_ragged")):
        lg = tiny.decode_tgt(torch.from_numpy(io[tgt_key]).cuda(), mem, memory_pad_mask=mask).cpu()
        ref = torch.from_numpy(io[out_key])
        diff = (lg - ref).abs().max().item()
        print(tgt_key, "logits max abs diff", diff)
        assert diff < LOGIT_TOL
        assert torch.equal(lg.argmax(-1), ref.argmax(-1))
    fwd = tiny(src.cuda(), torch.from_numpy(io["tgt_in"][:, :1]).cuda()).cpu()
    assert (fwd - torch.from_numpy(io["fwd_bos"])).abs().max().item() < LOGIT_TOL


def test_decode_with_shared_memory_rows(tiny):
    io = load_npz("tiny_model_io.npz")
    src = torch.from_numpy(io["src"])
    mem = torch.from_numpy(io["memory"]).cuda()
    mask = (src == 0).cuda()
    tgt = torch.from_numpy(io["tgt_in"]).cuda()
    rows = torch.tensor([3, 3, 0, 7, 7, 7], dtype=torch.int32)
    a = tiny.decode_tgt(tgt[rows.long()], mem[rows.long()], memory_pad_mask=mask[rows.long()])
    b = tiny.decode_tgt(tgt[rows.long()], mem, memory_pad_mask=mask, memory_row=rows.cuda())
    assert torch.equal(a, b)


def test_full_size_matches_reference(full):
    io = load_npz("full_model_io.npz")
    src, tgt = torch.from_numpy(io["src"]), torch.from_numpy(io["tgt"])
    mask = src == 0
    mem = full.encode_src(src.cuda(), mask.cuda()).cpu()
    ref_mem = torch.from_numpy(io["memory"])
    d_mem = (mem - ref_mem)[~mask].abs().max().item()
    lg = full.decode_tgt(tgt.cuda(), ref_mem.cuda(), memory_pad_mask=mask.cuda()).cpu()
    ref = torch.from_numpy(io["logits"])
    d_lg = (lg - ref).abs().max().item()
    print("full-size: memory diff", d_mem, "logits diff", d_lg, "logits absmax", ref.abs().max().item())
    assert d_mem < 1e-4
    assert d_lg < LOGIT_TOL
    assert torch.equal(lg.argmax(-1), ref.argmax(-1))


def test_make_drafts_matches_reference(tiny):
    gold = load_npz("drafts.npz")
    z = load_npz("fixture_tokens.npz")
    src, c = torch.from_numpy(z["src"]).cuda(), int(z["c_token"])
    for bsz in (1, 3, 10):
        for D in (1, 2, 3, 4, 5, 8, 10, 15, 25, 35, 50, 80, 100, 200):
            for N in (1, 2, 3, 5, 10, 15, 25, 35, 50, 80, 100, 200):
                got = tiny.make_drafts(src[:bsz], D, N, 1, 200, 2, 0, c).cpu().numpy()
                np.testing.assert_array_equal(got, gold[f"full_b{bsz}_d{D}_n{N}"], err_msg=f"b{bsz} d{D} n{N}")
    for D in (3, 5, 10, 17):
        for N in (1, 2, 3, 7, 23):
            np.testing.assert_array_equal(tiny.make_drafts(src[:, 1:], D, N, 1, 200, 2, 0, c).cpu().numpy(),
                                          gold[f"nobos_d{D}_n{N}"])
            np.testing.assert_array_equal(tiny.make_drafts(src[:, 1:], D, N, 5, 200, 2, 0, c).cpu().numpy(),
                                          gold[f"nobos_min5_d{D}_n{N}"])
    for D in (5, 10):
        np.testing.assert_array_equal(tiny.make_drafts(src, D + 1, src.shape[1] - 5, 5, 200, 2, 0, c).cpu().numpy(),
                                      gold[f"smartlib_d{D}"])
