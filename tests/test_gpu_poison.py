"""The row schedule on the 2-head test model with every fresh workspace filled with NaN bytes (TTX_POISON_WORKSPACES=1).
Regression test: the slot pool once left the cross K/V behind a slot's own source positions unwritten, and the attention
kernel of small models read them (0 x NaN -> NaN logits -> an arg-max sentinel used as a token id -> out-of-bounds
embedding read): an abort that showed up once in a few runs.  The switch is read when the library first allocates, so the
check runs in a child process."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

CHILD = r"""
import sys, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import translation_transformer_amd as tta
from util_models import tiny_state, fixture_tokens, PAD, BOS, EOS
st, cfg = tiny_state()
m = tta.NativeTransformer(st, cfg["num_heads"], 0, device=0)
src, _, c, _ = fixture_tokens()
batches = []
for lo, hi in ((0, 3), (3, 4), (4, 8), (8, 10), (0, 10), (5, 9), (6, 7)):
    sel = src[lo:hi]
    batches.append(sel[:, :int((sel != PAD).sum(1).max())].cuda())
for cap, fl in ((3, 2), (8, 1), (512, 4)):
    g = tta.TranslationInferenceGreedySpeculative(m, 150, 10, 3, PAD, BOS, EOS, c)
    out = g.generate_many(batches, in_flight=fl, reorder=True, group_size=cap)
    assert "device" in g.stats_total
    for b, o in zip(batches, out):
        ref = tta.TranslationInferenceGreedySpeculative(m, 150, 10, 3, PAD, BOS, EOS, c).generate(b)
        assert torch.equal(o, ref)
print("poisoned-workspace run ok")
"""


def test_row_schedule_with_poisoned_workspaces():
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, TTX_POISON_WORKSPACES="1")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": str(root), "tests": str(root / "tests")}], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "poisoned-workspace run ok" in r.stdout


CHILD_GROWTH = r"""
import sys, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import translation_transformer_amd as tta
from util_models import tiny_state, fixture_tokens, PAD, BOS, EOS
st, cfg = tiny_state()
m = tta.NativeTransformer(st, cfg["num_heads"], 0, device=0)
src, _, c, _ = fixture_tokens()
sel = src[:4]
b = sel[:, :int((sel != PAD).sum(1).max())].cuda()
g = tta.TranslationInferenceGreedySpeculative(m, 150, 10, 3, PAD, BOS, EOS, c)
first = g.generate(b)
second = g.generate(b)                      # second call of a shape captures the step graph
assert torch.equal(first, second)
# grow the SAME session's activation buffers through the model protocol (full-prefix decoder on many rows):
big = src[:10].repeat(24, 1)[:, :60].cuda()
mem = m.encode_src(big)
tgt = torch.full((big.shape[0], 150), 5, dtype=torch.int64, device="cuda"); tgt[:, 0] = BOS
logits = m.decode_tgt(tgt, mem, big == PAD)
assert torch.isfinite(logits).all()
# a couple of unrelated torch allocations that may land where the retired workspaces were
junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(8)]
third = g.generate(b)                       # must not replay a graph that points into the retired buffers
assert torch.equal(first, third), "tokens changed after the workspaces moved"
fourth = g.generate(b)
assert torch.equal(first, fourth)
print("graph-after-growth run ok")
"""


def test_graph_is_dropped_when_another_entry_point_grows_the_workspaces():
    """captured step graphs hold raw workspace pointers; encode_src / decode_tgt on the same session
    used to grow (move) those buffers without invalidating the graphs.  generate x2 (captures) -> decode_tgt with far
    more rows -> generate again must give the first call's tokens (NaN-poisoned fresh workspaces make a stale replay
    show up as different tokens or a fault)."""
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, TTX_POISON_WORKSPACES="1")
    r = subprocess.run([sys.executable, "-c", CHILD_GROWTH % {"root": str(root), "tests": str(root / "tests")}], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "graph-after-growth run ok" in r.stdout
