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
