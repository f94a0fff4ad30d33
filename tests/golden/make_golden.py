#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE implementation.

Runs ONLY in the build container, where /root/reference is mounted read-only.  It imports
the reference's hot-path modules (src/utils/drafting.py, src/decoding/*.py, src/model/modules.py,
src/model/embeddings.py, src/data_handling/tokenizer_*.py) exactly as SURVEY.md §8(c) describes
(stub parent packages so the Lightning-importing __init__ files are never executed), runs them on
fixed inputs and writes inputs + outputs as small .npz / .json fixtures.  Nothing produced here
contains reference source text: the fixtures are token ids, weights trained by this script and
output arrays.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [section ...]
Sections: tokens tiny model fullsize drafts greedy beam spec_greedy spec_beam helpers tiny66 spec_beam66  (default: all)
"""
import json
import os
import sys
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")
REF_SRC = REF / "src"

sys.dont_write_bytecode = True
sys.path.insert(0, str(REF_SRC))
for _name in ("model", "data_handling"):
    _m = types.ModuleType(_name)
    _m.__path__ = [str(REF_SRC / _name)]
    sys.modules[_name] = _m

from utils.drafting import make_drafts  # noqa: E402
from decoding.standard_decoding import TranslationInferenceGreedy, TranslationInferenceBeamSearch  # noqa: E402
from decoding.speculative_decoding import (  # noqa: E402
    TranslationInferenceGreedySpeculative,
    TranslationInferenceBeamSearchSpeculative,
    topk_in_each_group,
    mask_with_num_logits_according_nucleus,
)
from model.modules import VanillaTransformer  # noqa: E402
from data_handling.tokenizer_smiles import ChemSMILESTokenizer  # noqa: E402

torch.set_num_threads(8)
PAD, BOS, EOS = 0, 1, 2

TINY = dict(num_encoder_layers=2, num_decoder_layers=2, embedding_dim=64, num_heads=2,
            feedforward_dim=128)
FULL = dict(num_encoder_layers=4, num_decoder_layers=4, embedding_dim=256, num_heads=8,
            feedforward_dim=2048)


# --------------------------------------------------------------------------------------
def load_fixture_lines():
    src = [l.strip() for l in open(REF / "tests/product_prediction_src_test.txt")]
    tgt = [l.strip() for l in open(REF / "tests/product_prediction_tgt_test.txt")]
    return src, tgt


def pad_rows(rows, value=PAD):
    L = max(len(r) for r in rows)
    out = np.full((len(rows), L), value, dtype=np.int64)
    for i, r in enumerate(rows):
        out[i, :len(r)] = r
    return out


def section_tokens():
    """G5: vocabulary built by the reference tokenizer on the 20 fixture lines + encoded ids."""
    src, tgt = load_fixture_lines()
    tkz = ChemSMILESTokenizer()
    tkz.train_tokenizer(src + tgt)
    src_ids = [tkz.encode(s) for s in src]
    tgt_ids = [tkz.encode(t) for t in tgt]
    vocab = {str(k): v for k, v in tkz.decoder_dict.items()}
    (HERE / "fixture_vocab.json").write_text(json.dumps(vocab, sort_keys=True, indent=0))
    np.savez_compressed(HERE / "fixture_tokens.npz", src=pad_rows(src_ids), tgt=pad_rows(tgt_ids),
                        c_token=np.int64(tkz.encoder_dict["c"]), vocab_size=np.int64(tkz.n_tokens))
    # decode round trip (what PredictionWriter does with predictions)
    dec = [tkz.decode(np.array(t)) for t in tgt_ids]
    assert dec == tgt
    print("tokens: V =", tkz.n_tokens, "c =", tkz.encoder_dict["c"],
          "src lens", [len(s) for s in src_ids], "tgt lens", [len(t) for t in tgt_ids])


def fixture_tokens():
    z = np.load(HERE / "fixture_tokens.npz")
    return torch.from_numpy(z["src"]), torch.from_numpy(z["tgt"]), int(z["c_token"]), int(z["vocab_size"])


def build_ref_model(V, cfg):
    return VanillaTransformer(V, V, cfg["num_encoder_layers"], cfg["num_decoder_layers"],
                              cfg["embedding_dim"], cfg["num_heads"], cfg["feedforward_dim"],
                              0.0, "relu", True, PAD, PAD)


def state_to_npz(model, path):
    sd = {k: v.detach().cpu().numpy().astype(np.float32) for k, v in model.state_dict().items()}
    np.savez_compressed(path, **sd)


def load_state(model, path):
    z = np.load(path)
    model.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files})


def section_tiny():
    """Train the tiny reference model on the 10 fixture pairs (CE loss, mean, no ignore_index —
    src/model/lightning_model.py:68,139-160) until greedy decoding reproduces all ten targets."""
    src, tgt, c_tok, V = fixture_tokens()
    torch.manual_seed(123456)
    model = build_ref_model(V, TINY)
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    crit = torch.nn.CrossEntropyLoss(reduction="mean")
    model.train()
    for step in range(4000):
        logits = model(src, tgt[:, :-1])
        loss = crit(logits.reshape(-1, V), tgt[:, 1:].reshape(-1))
        opt.zero_grad()
        loss.backward()
        opt.step()
        if step % 100 == 0:
            print("tiny step", step, "loss", float(loss))
        if float(loss) < 2e-3:
            break
    model.eval()
    with torch.inference_mode():
        g = TranslationInferenceGreedy(model, 150, PAD, BOS, EOS).generate(src)
    ok = 0
    for i in range(src.size(0)):
        L = int((tgt[i] != PAD).sum())
        ok += int(torch.equal(g[i, 0, :L], tgt[i, :L]))
    print("tiny: final loss", float(loss), "steps", step, "greedy exact", ok, "/ 10")
    assert ok == 10
    state_to_npz(model, HERE / "tiny_weights.npz")
    (HERE / "tiny_config.json").write_text(json.dumps(dict(TINY, vocab_size=V, share_embeddings=True)))


def tiny_model():
    _, _, _, V = fixture_tokens()
    m = build_ref_model(V, TINY)
    load_state(m, HERE / "tiny_weights.npz")
    m.eval()
    return m


def section_model():
    """G2 (tiny): encode_src / decode_tgt / forward outputs of the reference model."""
    src, tgt, _, V = fixture_tokens()
    m = tiny_model()
    with torch.inference_mode():
        mask = src == PAD
        memory = m.encode_src(src, mask)
        logits = m.decode_tgt(tgt[:, :-1], memory, memory_pad_mask=mask)
        fwd = m(src, tgt[:, :1])
        # a second, ragged decoder input: prefix of varying length followed by PAD, as the
        # speculative loop feeds it (speculative_decoding.py:97-120)
        tgt2 = tgt[:, :24].clone()
        for i in range(tgt2.size(0)):
            tgt2[i, 8 + i:] = PAD
        logits2 = m.decode_tgt(tgt2, memory, memory_pad_mask=mask)
    np.savez_compressed(HERE / "tiny_model_io.npz", src=src.numpy(), tgt_in=tgt[:, :-1].numpy(),
                        memory=memory.numpy(), logits=logits.numpy(), fwd_bos=fwd.numpy(),
                        tgt_ragged=tgt2.numpy(), logits_ragged=logits2.numpy())
    print("model: memory", tuple(memory.shape), "logits", tuple(logits.shape))


def seeded_weights(shapes, seed):
    """Deterministic, RNG-library-independent weights: splitmix64 hash -> uniform(-a, a).
    Re-implemented identically in tests/util_weights.py; only this rule and the outputs are
    committed, the 46 MB of full-size weights are regenerated from it."""
    out = {}
    mask64 = (1 << 64) - 1
    ctr = np.uint64(seed)
    for name, shape in shapes:
        n = int(np.prod(shape))
        idx = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) + ctr
        z = idx
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
        u = (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # [0,1)
        fan_in = shape[-1] if len(shape) > 1 else None
        if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name.endswith("norm3.weight") \
                or name.endswith("norm.weight"):
            w = 1.0 + 0.2 * (u - 0.5)
        elif fan_in is None:
            w = 0.2 * (u - 0.5)
        else:
            a = 1.7 / np.sqrt(fan_in)
            w = a * (2.0 * u - 1.0)
        out[name] = w.astype(np.float32).reshape(shape)
        ctr = np.uint64((int(ctr) + 0x632BE59BD9B4E019 * (n + 1)) & mask64)
    return out


def section_fullsize():
    """G2 (full size, d=256/H=8/F=2048/4+4): weights from the seeded rule, reference outputs as slices."""
    V = 64
    torch.manual_seed(0)
    m = build_ref_model(V, FULL)
    names = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    with np.errstate(over="ignore"):
        w = seeded_weights(names, 20250725)
    w["tgt_token_featurizer.embedding.weight"] = w["src_token_featurizer.embedding.weight"]
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    m.eval()
    rng = np.random.default_rng(7)
    B, Ls, Lt = 6, 37, 29
    src = rng.integers(4, V, size=(B, Ls)).astype(np.int64)
    src[:, 0] = BOS
    lens = [37, 30, 21, 37, 12, 25]
    for i, L in enumerate(lens):
        src[i, L - 1] = EOS
        src[i, L:] = PAD
    tgt = rng.integers(4, V, size=(B, Lt)).astype(np.int64)
    tgt[:, 0] = BOS
    tl = [29, 20, 29, 11, 25, 17]
    for i, L in enumerate(tl):
        tgt[i, L:] = PAD
    with torch.inference_mode():
        s = torch.from_numpy(src)
        t = torch.from_numpy(tgt)
        mask = s == PAD
        memory = m.encode_src(s, mask)
        logits = m.decode_tgt(t, memory, memory_pad_mask=mask)
    np.savez_compressed(HERE / "full_model_io.npz", src=src, tgt=tgt, vocab_size=np.int64(V),
                        seed=np.int64(20250725),
                        memory=memory.numpy().astype(np.float32), logits=logits.numpy().astype(np.float32),
                        weight_names=np.array([n for n, _ in names]),
                        weight_checksum=np.float64(sum(float(np.abs(v).sum()) for v in w.values())))
    print("fullsize: logits absmax", float(logits.abs().max()), "memory absmax", float(memory.abs().max()))


def section_drafts():
    """G1: make_drafts on the fixture sources over the grid of the reference's tests/test_drafting.py
    (values, not only shapes), plus the two call shapes of the speculative generators."""
    src, _, c_tok, _ = fixture_tokens()
    lens = [1, 2, 3, 4, 5, 8, 10, 15, 25, 35, 50, 80, 100, 200]
    amts = [1, 2, 3, 5, 10, 15, 25, 35, 50, 80, 100, 200]
    out = {}
    for bsz in (1, 3, 10):
        for D in lens:
            for N in amts:
                d = make_drafts(src[:bsz], D, N, 1, 200, EOS, PAD, c_tok)
                assert tuple(d.shape) == (bsz, N, min(max(1, D), 200))
                out[f"full_b{bsz}_d{D}_n{N}"] = d.numpy().astype(np.int16)
    # greedy-speculative / all-drafts call shape: BOS dropped (speculative_decoding.py:64-73, :430)
    for D in (3, 5, 10, 17):
        for N in (1, 2, 3, 7, 23):
            d = make_drafts(src[:, 1:], D, N, 1, 200, EOS, PAD, c_tok)
            out[f"nobos_d{D}_n{N}"] = d.numpy().astype(np.int16)
            d = make_drafts(src[:, 1:], D, N, 5, 200, EOS, PAD, c_tok)
            out[f"nobos_min5_d{D}_n{N}"] = d.numpy().astype(np.int16)
    # smart-drafts library call shape (speculative_decoding.py:603-615)
    for D in (5, 10):
        d = make_drafts(src, D + 1, src.shape[1] - 5, 5, 200, EOS, PAD, c_tok)
        out[f"smartlib_d{D}"] = d.numpy().astype(np.int16)
    np.savez_compressed(HERE / "drafts.npz", **out)
    print("drafts:", len(out), "arrays")


def trim_np(t):
    return t.numpy().astype(np.int16)


def section_greedy():
    src, _, _, _ = fixture_tokens()
    m = tiny_model()
    out = {}
    with torch.inference_mode():
        for bsz in (1, 4, 10):
            for max_len in (150, 40):
                g = TranslationInferenceGreedy(m, max_len, PAD, BOS, EOS)
                toks = []
                for i in range(0, 10, bsz):
                    toks.append(g.generate(src[i:i + bsz]))
                w = max(t.size(2) for t in toks)
                arr = np.concatenate([np.pad(trim_np(t), ((0, 0), (0, 0), (0, w - t.size(2)))) for t in toks])
                out[f"b{bsz}_m{max_len}_tokens"] = arr
                out[f"b{bsz}_m{max_len}_calls"] = np.int64(g.model_calls_num)
    np.savez_compressed(HERE / "gen_greedy.npz", **out)
    print("greedy ok")


def section_beam():
    src, _, _, _ = fixture_tokens()
    m = tiny_model()
    out = {}
    with torch.inference_mode():
        for bsz, beam in ((1, 5), (4, 5), (10, 3), (5, 10)):
            g = TranslationInferenceBeamSearch(m, beam, 150, PAD, BOS, EOS)
            for bi, i in enumerate(range(0, 10, bsz)):
                t = g.generate(src[i:i + bsz])
                out[f"b{bsz}_k{beam}_batch{bi}"] = trim_np(t)
            out[f"b{bsz}_k{beam}_calls"] = np.int64(g.model_calls_num)
    np.savez_compressed(HERE / "gen_beam.npz", **out)
    print("beam ok")


def section_spec_greedy():
    src, _, c_tok, _ = fixture_tokens()
    m = tiny_model()
    out = {}
    with torch.inference_mode():
        for bsz in (1, 4, 10):
            for N in (1, 3, 7, 23):
                for D in (5, 10, 17):
                    g = TranslationInferenceGreedySpeculative(m, 150, D, N, PAD, BOS, EOS, c_tok)
                    toks = [g.generate(src[i:i + bsz]) for i in range(0, 10, bsz)]
                    out[f"b{bsz}_n{N}_d{D}_tokens"] = np.concatenate([trim_np(t) for t in toks])
                    out[f"b{bsz}_n{N}_d{D}_calls"] = np.int64(g.model_calls_num)
        # max_len small enough that some rows never finish (quirk: they stay all-PAD)
        for max_len in (30, 45):
            g = TranslationInferenceGreedySpeculative(m, max_len, 10, 3, PAD, BOS, EOS, c_tok)
            t = g.generate(src)
            out[f"short_m{max_len}_tokens"] = trim_np(t)
            out[f"short_m{max_len}_calls"] = np.int64(g.model_calls_num)
    np.savez_compressed(HERE / "gen_spec_greedy.npz", **out)
    print("spec greedy ok")


def section_spec_beam():
    """G3 (beam-speculative, both draft modes).  The reference loop does not terminate when a low-ranked
    candidate of the overfit tiny model keeps predicting PAD without ever reaching EOS (observed for
    fixture rows 1 and 7 at n_best >= 5: the same decoder input is re-submitted for ever), so each
    configuration names the fixture rows it uses and runs under a cap on decoder calls."""
    src, _, c_tok, V = fixture_tokens()
    m = tiny_model()
    real_decode = m.decode_tgt
    calls = [0]

    def capped(*a, **k):
        calls[0] += 1
        assert calls[0] < 400, "reference beam-speculative loop is not terminating on this input"
        return real_decode(*a, **k)

    m.decode_tgt = capped
    cases = [
        # rows, batch size, n_best, n_drafts, draft_len
        ([0, 2, 3, 4], 4, 5, 7, 10),
        ([5, 6, 8, 9], 4, 5, 3, 10),
        ([0, 1, 2, 3, 4, 5, 6, 7, 8, 9], 2, 3, 2, 5),
        ([0, 2, 3, 4, 5, 6, 8, 9], 8, 10, 2, 10),
        ([2, 4, 6, 9], 4, 5, 23, 17),
        ([6], 1, 5, 3, 10),
        ([0, 2, 3], 3, 2, 1, 3),
    ]
    out = {}
    with torch.inference_mode():
        for smart in (False, True):
            for ci, (rows, bsz, nbest, N, D) in enumerate(cases):
                g = TranslationInferenceBeamSearchSpeculative(
                    m, max_len=150, n_best=nbest, draft_len=D, n_drafts=N, vocab_size=V,
                    smart_drafts_mode=smart, pad_token=PAD, bos_token=BOS, eos_token=EOS, C_token=c_tok)
                key = f"smart{int(smart)}_case{ci}"
                out[f"{key}_rows"] = np.array(rows, dtype=np.int64)
                out[f"{key}_params"] = np.array([bsz, nbest, N, D], dtype=np.int64)
                nb = 0
                for bi, i in enumerate(range(0, len(rows), bsz)):
                    calls[0] = 0
                    sel = src[rows[i:i + bsz]]
                    # the loader pads every batch to its own longest row (seq2seq_wrappers.py:121-127)
                    width = int((sel != PAD).sum(1).max())
                    t = g.generate(sel[:, :width])
                    out[f"{key}_batch{bi}"] = trim_np(t)
                    nb += 1
                out[f"{key}_nbatches"] = np.int64(nb)
                out[f"{key}_calls"] = np.int64(g.model_calls_num)
                out[f"{key}_accepted"] = np.int64(g.accepted_tokens_num)
                out[f"{key}_produced"] = np.int64(g.produced_non_pad_tokens)
                print(key, rows, (bsz, nbest, N, D), "calls", g.model_calls_num, "acc", g.accepted_tokens_num,
                      g.produced_non_pad_tokens, flush=True)
    m.decode_tgt = real_decode
    np.savez_compressed(HERE / "gen_spec_beam.npz", **out)
    print("spec beam ok")


# --------------------------------------------------------------------------------------
# Config C4's shape in miniature: a 6+6-layer model (configs/cfg_standard_single_step_retrosyn.yaml:96-103) and the
# reference's beam-speculative generator at bs=8, n_best=10, n_drafts=2, draft_len=10, max_len=200
# (scripts/single_step_retrosynthesis.sh:166-174), both draft modes.
TINY66 = dict(num_encoder_layers=6, num_decoder_layers=6, embedding_dim=64, num_heads=2, feedforward_dim=128)


def section_tiny66():
    src, tgt, c_tok, V = fixture_tokens()
    torch.manual_seed(1)            # a seed under which the reference's beam-speculative loop terminates on 8 of the 10 rows
    torch.set_num_threads(2)
    model = build_ref_model(V, TINY66)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    crit = torch.nn.CrossEntropyLoss(reduction="mean")
    model.train()
    for step in range(6000):
        logits = model(src, tgt[:, :-1])
        loss = crit(logits.reshape(-1, V), tgt[:, 1:].reshape(-1))
        opt.zero_grad()
        loss.backward()
        opt.step()
        if step % 100 == 0:
            print("tiny66 step", step, "loss", float(loss), flush=True)
        if float(loss) < 2e-3:
            break
    model.eval()
    with torch.inference_mode():
        g = TranslationInferenceGreedy(model, 150, PAD, BOS, EOS).generate(src)
    ok = 0
    for i in range(src.size(0)):
        L = int((tgt[i] != PAD).sum())
        ok += int(torch.equal(g[i, 0, :L], tgt[i, :L]))
    print("tiny66: final loss", float(loss), "steps", step, "greedy exact", ok, "/ 10")
    torch.set_num_threads(8)
    assert ok == 10
    state_to_npz(model, HERE / "tiny66_weights.npz")
    (HERE / "tiny66_config.json").write_text(json.dumps(dict(TINY66, vocab_size=V, share_embeddings=True)))


def section_spec_beam66():
    src, _, c_tok, V = fixture_tokens()
    m = build_ref_model(V, TINY66)
    load_state(m, HERE / "tiny66_weights.npz")
    m.eval()
    real_decode = m.decode_tgt
    calls = [0]

    class NotTerminating(Exception):
        pass

    def capped(*a, **k):
        calls[0] += 1
        if calls[0] >= 400:
            raise NotTerminating()
        return real_decode(*a, **k)

    m.decode_tgt = capped

    def make(smart, nbest, N, D):
        return TranslationInferenceBeamSearchSpeculative(
            m, max_len=200, n_best=nbest, draft_len=D, n_drafts=N, vocab_size=V, smart_drafts_mode=smart,
            pad_token=PAD, bos_token=BOS, eos_token=EOS, C_token=c_tok)

    out = {}
    with torch.inference_mode():
        # rows on which the reference loop terminates in both modes at n_best = 10 (see section_spec_beam's docstring)
        good = []
        for r in range(src.size(0)):
            ok = True
            for smart in (False, True):
                calls[0] = 0
                sel = src[r:r + 1]
                try:
                    make(smart, 10, 2, 10).generate(sel[:, :int((sel != PAD).sum())])
                except NotTerminating:
                    ok = False
            print("row", r, "terminates" if ok else "does NOT terminate", flush=True)
            if ok:
                good.append(r)
        cases = [(good[:8], 8, 10, 2, 10), (good[:8], 4, 10, 2, 11), (good[:6], 3, 5, 7, 10)]
        for smart in (False, True):
            for ci, (rows, bsz, nbest, N, D) in enumerate(cases):
                g = make(smart, nbest, N, D)
                key = f"smart{int(smart)}_case{ci}"
                out[f"{key}_rows"] = np.array(rows, dtype=np.int64)
                out[f"{key}_params"] = np.array([bsz, nbest, N, D], dtype=np.int64)
                nb = 0
                for bi, i in enumerate(range(0, len(rows), bsz)):
                    calls[0] = 0
                    sel = src[rows[i:i + bsz]]
                    width = int((sel != PAD).sum(1).max())
                    out[f"{key}_batch{bi}"] = trim_np(g.generate(sel[:, :width]))
                    nb += 1
                out[f"{key}_nbatches"] = np.int64(nb)
                out[f"{key}_calls"] = np.int64(g.model_calls_num)
                out[f"{key}_accepted"] = np.int64(g.accepted_tokens_num)
                out[f"{key}_produced"] = np.int64(g.produced_non_pad_tokens)
                out[f"{key}_lines"] = np.int64(g.model_input_lines_num)
                print(key, rows, (bsz, nbest, N, D), "calls", g.model_calls_num, "acc", g.accepted_tokens_num,
                      g.produced_non_pad_tokens, flush=True)
    m.decode_tgt = real_decode
    np.savez_compressed(HERE / "gen_spec_beam66.npz", **out)
    print("spec beam 6+6 ok")


def section_helpers():
    """G4: nucleus masking and per-group top-k on fixed tensors."""
    rng = np.random.default_rng(11)
    out = {}
    logits = (rng.standard_normal((6, 5, 30)) * 3.0).astype(np.float32)
    logits[0, 0, 3] = 35.0  # a 'finished row' style distribution
    out["nuc_in"] = logits
    for nucleus, nbest, num, tag in ((0.9975, 5, "-inf", "a"), (20.0, 5, 0.0, "b"), (0.9975, 10, "-inf", "c"),
                                     (0.5, 3, 0.0, "d")):
        r = mask_with_num_logits_according_nucleus(torch.from_numpy(logits.copy()), nucleus, nbest, num)
        out[f"nuc_out_{tag}"] = r.numpy()
    score = rng.standard_normal((23, 1)).astype(np.float32)
    lens = np.array([5, 7, 4, 7], dtype=np.int64)
    s, idx = topk_in_each_group(torch.from_numpy(score.copy()), torch.from_numpy(lens), 3, pad=-float("inf"))
    out["topk_score_in"] = score
    out["topk_lens"] = lens
    out["topk_score_out"] = s.numpy()
    out["topk_idx_out"] = idx.numpy()
    lens2 = np.array([6, 6, 6], dtype=np.int64)
    s, idx = topk_in_each_group(torch.from_numpy(score[:18].copy()), torch.from_numpy(lens2), 2, pad=-float("inf"))
    out["topk2_lens"] = lens2
    out["topk2_score_out"] = s.numpy()
    out["topk2_idx_out"] = idx.numpy()
    np.savez_compressed(HERE / "helpers.npz", **out)
    print("helpers ok")


def section_tokenizer():
    """G5 extended: the reference tokenizer (regex split, encode with BOS/EOS/UNK, decode to first EOS) on the fixture
    lines and on synthetic SMILES-like strings that exercise every alternative of its regex (tokenizer_smiles.py:8)."""
    from data_handling.tokenizer_smiles import SimpleSmilesTokenizer
    src, tgt = load_fixture_lines()
    tkz = ChemSMILESTokenizer()
    tkz.train_tokenizer(src + tgt)
    rng = np.random.default_rng(2024)
    atoms = ["C", "N", "O", "S", "P", "F", "I", "B", "Br", "Cl", "b", "c", "n", "o", "s", "p", "[Na+]", "[O-]", "[N+]", "[nH]",
             "[C@@H]", "[13CH3]", "(", ")", ".", "=", "#", "-", "+", "\\", "/", ":", "~", "@", "?", ">", "*", "$", "%10", "%99",
             "1", "2", "9", "0"]
    junk = ["X", "x", "Z", "[", "]", "%", "%1", "[]", "[ab", " ", "r", "l", "é", "H", "Si", "Se"]
    lines = list(src) + list(tgt)
    for _ in range(400):
        n = int(rng.integers(1, 40))
        parts = []
        for _ in range(n):
            pool = junk if rng.random() < 0.12 else atoms
            parts.append(pool[int(rng.integers(0, len(pool)))])
        lines.append("".join(parts))
    lines += ["", "[", "]", "[[C]]", "Brr", "ClCl", "BrCl", "%123", "%1a", "C%12C", "\\\\", "C1=CC=CC=C1", "[Na+].[Cl-]"]
    pieces = [SimpleSmilesTokenizer.split_into_tokens(l) for l in lines]
    ids = [tkz.encode(l) for l in lines]
    dec = [tkz.decode(np.array(i)) for i in ids]
    dec_raw = [tkz.decode(np.array(i), skip_service_tokens=False) for i in ids]
    (HERE / "tokenizer_cases.json").write_text(json.dumps(
        {"vocab": {str(k): v for k, v in tkz.decoder_dict.items()}, "lines": lines, "pieces": pieces, "ids": ids,
         "decoded": dec, "decoded_with_service": dec_raw}, ensure_ascii=False))
    print("tokenizer:", len(lines), "lines,", sum(len(p) for p in pieces), "tokens")


SECTIONS = dict(tokenizer=section_tokenizer, tokens=section_tokens, tiny=section_tiny, model=section_model, fullsize=section_fullsize,
                drafts=section_drafts, greedy=section_greedy, beam=section_beam,
                spec_greedy=section_spec_greedy, spec_beam=section_spec_beam, helpers=section_helpers,
                tiny66=section_tiny66, spec_beam66=section_spec_beam66)

if __name__ == "__main__":
    todo = sys.argv[1:] or list(SECTIONS)
    for name in todo:
        SECTIONS[name]()
