"""csrc/ttx_select.h (which equal maximum `topk(1)` returns) compiled for the host and checked against torch.topk on the
CPU — the reference's own tie-breaking (speculative_decoding.py:553, :779-784 via :225)."""
import ctypes
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "translation-transformer_amd" / "csrc"


@pytest.fixture(scope="module")
def sel(tmp_path_factory):
    out = tmp_path_factory.mktemp("sel") / "libsel.so"
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", f"-I{CSRC}", "-o", str(out), str(ROOT / "tests" / "host" / "select_host.cpp")],
                   check=True)
    lib = ctypes.CDLL(str(out))
    lib.ttx_host_topk1_index.restype = ctypes.c_int
    lib.ttx_host_topk1_index.argtypes = [ctypes.c_void_p, ctypes.c_int]
    return lib


def test_topk1_ties_match_torch_cpu(sel):
    rng = np.random.default_rng(7)
    checked = 0
    for n in list(range(1, 70)) + [100, 127, 128, 130]:
        for hi in (1, 2, 3, 5, 11):                    # few distinct values: ties everywhere
            x = rng.integers(0, hi + 1, size=(400, n)).astype(np.int64)
            ref = torch.from_numpy(x).topk(1, dim=-1).indices[:, 0].numpy()
            for r in range(x.shape[0]):
                got = sel.ttx_host_topk1_index(x[r].ctypes.data, n)
                assert got == ref[r], (n, x[r].tolist(), got, int(ref[r]))
                checked += 1
    assert checked > 100000


def test_topk1_padded_groups_match_topk_in_each_group(sel):
    """smart-drafts mode: group g has len_g scores, the table is padded with -1 to the longest group before topk(1)."""
    rng = np.random.default_rng(11)
    for width in (2, 3, 4, 7, 10, 23):
        for _ in range(300):
            ln = int(rng.integers(1, width + 1))
            row = np.full(width, -1, dtype=np.int64)
            row[:ln] = rng.integers(0, 6, size=ln)
            ref = int(torch.from_numpy(row)[None].topk(1, dim=-1).indices[0, 0])
            assert sel.ttx_host_topk1_index(row.ctypes.data, width) == ref
            assert ref < ln
