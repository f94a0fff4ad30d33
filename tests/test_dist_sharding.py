"""The N>1 data path on CPU: world_size-2 gloo processes shard a test set, 'decode' their shards, broadcast
weights once and gather predictions once — the same calls bench.py makes over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import translation_transformer_amd  # noqa: F401
from translation_transformer_amd.dist import shard_bounds, broadcast_state_dict, gather_predictions, sum_counters


def test_shard_bounds_cover_without_overlap():
    for n in (0, 1, 7, 32, 33, 500, 40000):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                lo, hi = shard_bounds(n, r, world)
                assert 0 <= lo <= hi <= n
                seen.extend(range(lo, hi))
            assert seen == list(range(n))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_items, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        state = None
        if rank == 0:
            g = torch.Generator().manual_seed(5)
            state = {"a.weight": torch.randn(7, 3, generator=g), "b.bias": torch.randn(11, generator=g)}
        state = broadcast_state_dict(state, "cpu", dist)
        lo, hi = shard_bounds(n_items, rank, world)
        # fake decoder: prediction of item i is a row that encodes i; ragged widths per rank
        width = 5 + rank
        local = torch.zeros((hi - lo, 2, width), dtype=torch.int64)
        for j, i in enumerate(range(lo, hi)):
            local[j, 0, :3] = torch.tensor([1, i % 100 + 4, 2])
            local[j, 1, :2] = torch.tensor([1, 2])
        out = gather_predictions(local, n_items, dist)
        tot = sum_counters({"calls": 10 * (rank + 1), "acc": rank}, "cpu", dist)
        if rank == 0:
            q.put((float(state["a.weight"].sum()), float(state["b.bias"].sum()), out.numpy(), tot))
        else:
            assert out is None
            q.put((float(state["a.weight"].sum()), float(state["b.bias"].sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [9, 10])
def test_two_rank_broadcast_and_gather(n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = [r for r in res if len(r) == 4][0]
    other = [r for r in res if len(r) == 2][0]
    assert full[0] == other[0] and full[1] == other[1]          # every rank holds rank 0's weights
    out = full[2]
    assert out.shape == (n_items, 2, 6)
    for i in range(n_items):
        assert out[i, 0, :3].tolist() == [1, i % 100 + 4, 2]
    assert full[3]["calls"] == 30 and full[3]["acc"] == 1
