"""Cell sharding across ranks: one process per GPU, cells are independent units.

The ascending-id cell list of an image is cut into contiguous, balanced shards (rank order = CSV order).  The only exchange
on the data path is one all-gather of the per-cell rows each rank produced (softmax probabilities, <= 33 floats per cell;
intensity means): ``torch.distributed`` backend "nccl" is RCCL over xGMI on MI355X, "gloo" is used by the CPU tests.
"""
from __future__ import annotations

from typing import Tuple

import torch


def world() -> Tuple[int, int]:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous balanced split of range(n): sizes differ by at most one, concatenation in rank order restores the order."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world size")
    return (n * rank) // world_size, (n * (rank + 1)) // world_size


def owns_image(i: int, rank: int, world_size: int) -> bool:
    """Tile-per-rank mode (BASELINE config 5, reference main.py:39-52 ``batch_run``): image i of the batch CSV belongs to rank
    i mod world_size.  Whole images are independent units: replicas only, no collective on the data path."""
    return i % world_size == rank


def tile_mode(n_images: int, world_size: int, env=None) -> bool:
    """Which sharding a multi-rank run uses: whole IMAGES per rank when the batch has at least one image per rank (every rank then
    normalises, crops, classifies and writes its own tiles, nothing is exchanged), CELLS of every image otherwise (contiguous shards + one
    all-gather per image).  RIBCA_TILE_MODE=0 / 1 overrides the rule."""
    if world_size <= 1:
        return False
    if env is not None and env != "":
        return env == "1"
    return n_images >= world_size


def all_reduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    """Sum of a small host-side statistic over the ranks (the cell-type co-occurrence counts of an integrated neighbourhood analysis in
    tile-per-rank mode: T x T numbers, downstream of the data path)."""
    import torch.distributed as dist
    if world()[1] == 1:
        return t
    if t.is_cuda and dist.get_backend(group) == "gloo":
        return all_reduce_sum(t.cpu(), group).to(t.device)
    if not t.is_cuda and dist.get_backend(group) == "nccl":      # RCCL reduces device buffers only
        return all_reduce_sum(t.cuda(), group).cpu()
    out = t.clone()
    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    return out


def all_reduce_min_int(v: int, group=None) -> int:
    """Smallest of one integer per rank (control plane: keeps data-dependent branches of a multi-rank pipeline in step)."""
    import torch.distributed as dist
    if world()[1] == 1:
        return int(v)
    t = torch.tensor([int(v)], dtype=torch.int64)
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t.item())


def all_gather_planes(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Channel-sharded whole-image normalisation (reference preprocess.py:214-239 treats every channel on its own): rank r holds the
    finished fp32 planes ``shard_bounds(C, r, world)`` of a (C, H, W) image; one all-gather of the planes gives every rank the image.  The
    same padded fixed-size collective as all_gather_rows (a plane is a row of H * W values)."""
    return all_gather_rows(local, n_total, group)


def all_gather_rows(local: torch.Tensor, n_total: int, group=None, force_collective: bool = False) -> torch.Tensor:
    """Reassemble a (n_total, K) tensor from each rank's contiguous (n_local, K) shard (``shard_bounds`` layout).
    Shards are padded to the largest shard so a single fixed-size all_gather (one RCCL call) is enough.
    ``force_collective`` issues the collective even in a group of one rank (the single-GPU RCCL test: same call, same buffers)."""
    import torch.distributed as dist
    rank, ws = world()
    if ws == 1 and not (force_collective and dist.is_initialized()):
        assert local.shape[0] == n_total
        return local
    if local.is_cuda and dist.get_backend(group) == "gloo":     # rehearsal / CPU-collective runs: stage through host memory
        return all_gather_rows(local.cpu(), n_total, group).to(local.device)
    k = local.shape[1:]
    cap = max(shard_bounds(n_total, r, ws)[1] - shard_bounds(n_total, r, ws)[0] for r in range(ws))
    send = torch.zeros((cap,) + tuple(k), dtype=local.dtype, device=local.device)
    send[:local.shape[0]] = local
    recv = torch.empty((ws * cap,) + tuple(k), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    parts = []
    for r in range(ws):
        lo, hi = shard_bounds(n_total, r, ws)
        parts.append(recv[r * cap:r * cap + (hi - lo)])
    return torch.cat(parts, dim=0)
