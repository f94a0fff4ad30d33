"""Multi-GPU layer: the test set shards embarrassingly over ranks, one process per GPU.
Two collectives per run and none per step:  C1 one broadcast of the packed fp32 weights from rank 0,
C2 one gather of the predictions to rank 0.  Backend "nccl" is RCCL over xGMI on ROCm; the same code runs
on "gloo" for the CPU tests (tests/test_dist_sharding.py)."""
from __future__ import annotations

import math

import numpy as np
import torch


def shard_bounds(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous slice [lo, hi) of an unshuffled test set for `rank` (no duplication, unlike Lightning's
    DistributedSampler which pads shards to equal length)."""
    per = math.ceil(n_items / world) if world > 0 else n_items
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


def _wire_device(device, dist):
    """Collectives run on the GPU under RCCL ("nccl") and on host memory under gloo (CPU tests, rehearsals)."""
    return torch.device("cpu") if dist.get_backend() == "gloo" else torch.device(device)


def broadcast_state_dict(state: dict | None, device: torch.device | str, dist=None, src: int = 0) -> dict:
    """C1.  Rank `src` passes its state dict (name -> tensor); every rank returns the same dict (CPU fp32).
    One broadcast of names/shapes (tiny, object) + ONE broadcast of the flat fp32 blob (46 MB for the MIT model)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return {k: torch.as_tensor(v).float().cpu() for k, v in state.items()}
    rank = dist.get_rank()
    device = _wire_device(device, dist)
    meta = [None]
    if rank == src:
        names = list(state.keys())
        meta = [[(k, tuple(state[k].shape)) for k in names]]
        flat = torch.cat([torch.as_tensor(state[k]).reshape(-1).float() for k in names]).to(device)
    dist.broadcast_object_list(meta, src=src)
    total = sum(int(np.prod(s)) for _, s in meta[0])
    if rank != src:
        flat = torch.empty(total, dtype=torch.float32, device=device)
    dist.broadcast(flat, src=src)
    host = flat.cpu()
    out, off = {}, 0
    for k, s in meta[0]:
        n = int(np.prod(s))
        out[k] = host[off:off + n].reshape(s)
        off += n
    return out


def broadcast_model(state: dict | None, num_heads: int, pad_token_idx: int, device, dist=None, src: int = 0):
    """C1 as SURVEY.md §8(e) words it: rank `src` builds the model from its state dict (weights packed into one HBM blob),
    every other rank creates an EMPTY model of the same shape (ttx_model_create_empty) and receives the blob with ONE
    RCCL broadcast straight into HBM (ttx_model_blob) — no host staging, no per-tensor messages.  Returns the
    NativeTransformer of this rank.  Under gloo (CPU tests, several ranks sharing one GPU) the weights travel as a host
    state dict instead (broadcast_state_dict) and every rank uploads them."""
    from .model import NativeTransformer, shape_of_state
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return NativeTransformer(state, num_heads, pad_token_idx, device=device)
    if dist.get_backend() == "gloo":
        sd = broadcast_state_dict(state, device, dist, src)
        return NativeTransformer(sd, num_heads, pad_token_idx, device=device)
    rank = dist.get_rank()
    meta = [shape_of_state(state) if rank == src else None]
    dist.broadcast_object_list(meta, src=src)
    model = NativeTransformer(state if rank == src else None, num_heads, pad_token_idx, device=device, shape=meta[0])
    blob = model.blob_tensor()
    torch.cuda.synchronize(blob.device)           # rank src: the upload of the blob has completed
    dist.broadcast(blob, src=src)
    torch.cuda.synchronize(blob.device)
    return model


def gather_predictions(local: torch.Tensor, n_items: int, dist=None, dst: int = 0, pad_value: int = 0):
    """C2.  `local` is this rank's [n_local, N, L] integer predictions for its shard_bounds slice; returns the
    [n_items, N, Lmax] tensor in original order on rank `dst` (None elsewhere).  Token ids travel as int32."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    home = local.device
    local = local.to(_wire_device(home, dist))
    per = math.ceil(n_items / world)
    dt = torch.int32            # token ids; every rank must agree on the wire type
    width = torch.tensor([local.shape[2] if local.numel() else 0], dtype=torch.int64, device=local.device)
    dist.all_reduce(width, op=dist.ReduceOp.MAX)
    W = int(width.item())
    buf = torch.full((per, local.shape[1], W), pad_value, dtype=dt, device=local.device)
    buf[:local.shape[0], :, :local.shape[2]] = local.to(dt)
    # all_gather rather than gather: supported by every RCCL/NCCL/gloo build; the extra copies are a few MB once per run
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    if rank != dst:
        return None
    out = []
    for r, p in enumerate(parts):
        lo, hi = shard_bounds(n_items, r, world)
        out.append(p[:hi - lo])
    return torch.cat(out).to(torch.int64).to(home)


def sum_counters(values: dict, device, dist=None) -> dict:
    """Additive per-rank counters (model calls, accepted tokens, ...) summed over ranks."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(values)
    keys = sorted(values)
    t = torch.tensor([float(values[k]) for k in keys], dtype=torch.float64, device=_wire_device(device, dist))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return {k: t[i].item() for i, k in enumerate(keys)}
