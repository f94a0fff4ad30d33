"""The opt-in FFN experiment (TTX_FFN_BF16X6=1: every product of the two FFN GEMMs formed from six bf16 MFMA partial products
with fp32 accumulation, csrc/ttx_kernels.hip.h b6_split / b6_mma): the full-size tests (reference logits, oracle tokens of the greedy, greedy-speculative and beam generators, the
KV-cached step logits, the slot-pool schedule) must stay green under it.  The switch is read when a session is created, so the selected tests run in a child pytest process.
(The whole `-m gpu` suite passes under the switch as well — gpurun_out/gputest_b6.log of round 2; this keeps a fast subset.)"""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_full_size_tests_stay_green_under_bf16x6_ffn():
    if os.environ.get("TTX_FFN_BF16X6") == "1":
        pytest.skip("already running under the switch")
    env = dict(os.environ, TTX_FFN_BF16X6="1")
    # the switch only applies where K per split is a multiple of 256 and one dimension is the FFN width, i.e. at the real layer
    # sizes (d = 256, FFN 2048): the full-size tests
    sel = ["tests/test_gpu_model.py::test_full_size_matches_reference",
           "tests/test_gpu_generators.py::test_full_size_greedy_speculative_matches_oracle",
           "tests/test_gpu_generators.py::test_full_size_greedy_and_beam_match_oracle",
           "tests/test_gpu_generators.py::test_kv_cached_step_logits_match_full_prefix_oracle",
           "tests/test_gpu_generators.py::test_full_size_row_schedule_pool_equals_per_batch_and_oracle"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", *sel], env=env, cwd=str(ROOT), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
