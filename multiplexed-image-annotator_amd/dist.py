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
