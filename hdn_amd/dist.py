"""Multi-GPU: independent template/search pairs are sharded across ranks; the only exchange is one
all-gather of the predicted corner offsets x[B_local, 8].

One process per GPU, torch.distributed with backend "nccl" (= RCCL over xGMI on ROCm); "gloo" for the
CPU tests.  The payload is 2 KB per rank at B=512 / world 8: latency-bound, one collective, no bucketing.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(n_pairs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of pairs owned by `rank`; the first n_pairs % world ranks own one extra."""
    if world <= 0 or not (0 <= rank < world) or n_pairs < 0:
        raise ValueError(f"bad shard request n_pairs={n_pairs} rank={rank} world={world}")
    q, r = divmod(n_pairs, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def all_gather_offsets(x_local: torch.Tensor, n_pairs: int = None, group=None) -> torch.Tensor:
    """Gather every rank's [B_local, 8] offsets into [n_pairs, 8] on every rank, in pair order.

    Equal shards use one all_gather_into_tensor; ragged shards (n_pairs % world != 0) pad to the
    largest shard so that it is still a single collective.
    """
    if x_local.dim() != 2:
        raise ValueError("x_local must be [B_local, D]")
    if not (dist.is_available() and dist.is_initialized()):
        return x_local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return x_local
    D = x_local.shape[1]
    if n_pairs is None:
        n_pairs = x_local.shape[0] * world
    sizes = [shard_range(n_pairs, r, world) for r in range(world)]
    mine = sizes[rank][1] - sizes[rank][0]
    if x_local.shape[0] != mine:
        raise ValueError(f"rank {rank} holds {x_local.shape[0]} pairs, shard_range says {mine}")
    cap = max(e - s for s, e in sizes)
    buf = x_local.contiguous()
    if mine != cap:
        buf = torch.cat([buf, buf.new_zeros((cap - mine, D))], dim=0)
    out = buf.new_empty((world * cap, D))
    if buf.is_cuda and dist.get_backend(group) == "gloo":
        # gloo has no device all-gather: stage through the host (CPU tests / single-GPU dry runs only; on the GPUs the
        # backend is nccl = RCCL and the collective runs on device memory)
        host = out.cpu()
        dist.all_gather_into_tensor(host, buf.cpu(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, buf, group=group)
    if n_pairs == world * cap:
        return out
    return torch.cat([out[r * cap: r * cap + (e - s)] for r, (s, e) in enumerate(sizes)], dim=0)


def sharded_offsets(net, data: dict, group=None) -> torch.Tensor:
    """Run the homography head on this rank's shard of `data` (dict of [B, ...] tensors, all ranks hold
    the same global batch) and return the gathered [B, 8] corner offsets."""
    from .homo_model import homo_stages

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = data["org_imgs"].shape[0]
    s, e = shard_range(B, rank, world)
    local = {k: v[s:e].contiguous() for k, v in data.items() if k in ("org_imgs", "input_tensors", "h4p", "patch_indices")}
    x = homo_stages(net, local)["x"]
    return all_gather_offsets(x, B, group)
