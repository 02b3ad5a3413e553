"""Multi-GPU: independent template/search pairs are sharded across ranks; the only exchange is one
all-gather of the predicted corner offsets x[B_local, 8].

One process per GPU, torch.distributed with backend "nccl" (= RCCL over xGMI on ROCm); "gloo" for the
CPU tests.  The payload is 2 KB per rank at B=512 / world 8: latency-bound, one collective, no bucketing.
"""
from __future__ import annotations

import ctypes
from typing import Tuple

import torch
import torch.distributed as dist

from . import _lib


def shard_range(n_pairs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of pairs owned by `rank`; the first n_pairs % world ranks own one extra."""
    if world <= 0 or not (0 <= rank < world) or n_pairs < 0:
        raise ValueError(f"bad shard request n_pairs={n_pairs} rank={rank} world={world}")
    q, r = divmod(n_pairs, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


class RcclComm:
    """An RCCL communicator owned through the C ABI (hdn_rccl_comm_create / hdn_allgather_offsets, include/hdn_hip.h):
    the collective is then a library call on device pointers, not a torch.distributed op.

    RcclComm.from_process_group(device) bootstraps over an initialised torch.distributed group (any backend: the
    128-byte unique id is broadcast as an object); RcclComm(world, rank, uid) over anything else.  The communicator
    binds to `device` (default: the current one)."""

    def __init__(self, world: int, rank: int, uid: bytes, device=None):
        if len(uid) != 128:
            raise ValueError("an RCCL unique id is 128 bytes")
        self.world, self.rank = int(world), int(rank)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        h = ctypes.c_void_p()
        with _lib.device_guard(self.device):
            rc = _lib.load().hdn_rccl_comm_create(ctypes.byref(h), self.world, self.rank, ctypes.c_char_p(uid))
        _lib.check(rc, "hdn_rccl_comm_create")
        self._h = h

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.load().hdn_rccl_unique_id(buf), "hdn_rccl_unique_id")
        return buf.raw

    @classmethod
    def from_process_group(cls, device=None, group=None):
        if dist.is_available() and dist.is_initialized():
            world, rank = dist.get_world_size(group), dist.get_rank(group)
            box = [cls.unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        else:
            world, rank, box = 1, 0, [cls.unique_id()]
        return cls(world, rank, box[0], device)

    def all_gather(self, local: torch.Tensor) -> torch.Tensor:
        """local [Bl, 8] on this communicator's device -> [world * Bl, 8]; asynchronous on torch's current stream."""
        dev = _lib.require_device(local)
        if dev != self.device:
            raise _lib.HdnHipError(f"communicator bound to {self.device}, tensor on {dev}")
        if local.dim() != 2 or local.shape[1] != 8 or local.shape[0] == 0:
            raise ValueError(f"expected [B_local, 8] corner offsets, got {tuple(local.shape)}")
        if self._h is None:
            raise _lib.HdnHipError("communicator destroyed")
        loc = local.detach().contiguous()
        out = torch.empty((self.world * loc.shape[0], 8), dtype=torch.float32, device=dev)
        with _lib.device_guard(dev):
            rc = _lib.load().hdn_allgather_offsets(_lib.ptr(loc), _lib.ptr(out), loc.shape[0], self._h, _lib.stream_ptr(dev))
        _lib.check(rc, "hdn_allgather_offsets")
        return out

    def destroy(self):
        if self._h is not None:
            h, self._h = self._h, None
            with _lib.device_guard(self.device):
                _lib.check(_lib.load().hdn_rccl_comm_destroy(h), "hdn_rccl_comm_destroy")


class OneShotGather:
    """The same exchange without RCCL: a direct-write gather through hipIpc-mapped windows (hdn_gather_* of the C ABI,
    include/hdn_hip.h) — one kernel per rank, one xGMI hop, no host state per call (capturable in a hipGraph).  For the 2 KB per
    rank of this path a ring's hops and proxy hand-offs are all latency; SURVEY.md §5 asks for exactly this form.

    OneShotGather.from_process_group(max_rows, device) exchanges the 64-byte window handles over an initialised
    torch.distributed group (any backend); OneShotGather(world, rank, max_rows, exchange=...) over anything else, where
    exchange(my_handle: bytes) -> list of every rank's handle in rank order.  Drop-in for RcclComm in all_gather_offsets /
    sharded_offsets (`comm=`).  Needs every pair of ranks to have peer access (one xGMI node)."""

    def __init__(self, world: int, rank: int, max_rows: int, exchange, device=None):
        self.world, self.rank, self.max_rows = int(world), int(rank), int(max_rows)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        h = ctypes.c_void_p()
        lib = _lib.load()
        with _lib.device_guard(self.device):
            _lib.check(lib.hdn_gather_create(ctypes.byref(h), self.world, self.rank, self.max_rows * 32), "hdn_gather_create")
            self._h = h
            try:
                mine = ctypes.create_string_buffer(64)
                _lib.check(lib.hdn_gather_handle(h, mine), "hdn_gather_handle")
                handles = exchange(mine.raw)
                if len(handles) != self.world or any(len(x) != 64 for x in handles):
                    raise ValueError("exchange() must return one 64-byte handle per rank")
                _lib.check(lib.hdn_gather_connect(h, ctypes.c_char_p(b"".join(handles))), "hdn_gather_connect")
            except Exception:
                self._h = None                     # (the window is freed; peers that already mapped it must not use it)
                lib.hdn_gather_destroy(h)
                raise

    @classmethod
    def from_process_group(cls, max_rows: int, device=None, group=None):
        if dist.is_available() and dist.is_initialized():
            world, rank = dist.get_world_size(group), dist.get_rank(group)

            def exchange(mine):
                box = [None] * world
                dist.all_gather_object(box, mine, group=group)
                return box
        else:
            world, rank, exchange = 1, 0, (lambda mine: [mine])
        return cls(world, rank, max_rows, exchange, device)

    def all_gather(self, local: torch.Tensor) -> torch.Tensor:
        """local [Bl, 8] on this object's device -> [world * Bl, 8]; asynchronous on torch's current stream."""
        dev = _lib.require_device(local)
        if dev != self.device:
            raise _lib.HdnHipError(f"gather window bound to {self.device}, tensor on {dev}")
        if local.dim() != 2 or local.shape[1] != 8 or local.shape[0] == 0 or local.shape[0] > self.max_rows:
            raise ValueError(f"expected [1..{self.max_rows}, 8] corner offsets, got {tuple(local.shape)}")
        if self._h is None:
            raise _lib.HdnHipError("gather window destroyed")
        loc = local.detach().to(torch.float32).contiguous()
        out = torch.empty((self.world * loc.shape[0], 8), dtype=torch.float32, device=dev)
        with _lib.device_guard(dev):
            rc = _lib.load().hdn_gather_offsets_oneshot(self._h, _lib.ptr(loc), _lib.ptr(out), loc.shape[0], _lib.stream_ptr(dev))
        _lib.check(rc, "hdn_gather_offsets_oneshot")
        return out

    def status(self) -> int:
        """0, or non-zero once a call gave up waiting for a peer (read it after the stream has drained)."""
        return int(_lib.load().hdn_gather_status(self._h)) if self._h is not None else 0

    def destroy(self):
        if self._h is not None:
            h, self._h = self._h, None
            with _lib.device_guard(self.device):
                _lib.check(_lib.load().hdn_gather_destroy(h), "hdn_gather_destroy")


def all_gather_offsets(x_local: torch.Tensor, n_pairs: int = None, group=None, comm=None,
                       always_collective: bool = False) -> torch.Tensor:
    """Gather every rank's [B_local, 8] offsets into [n_pairs, 8] on every rank, in pair order.

    Equal shards are one all-gather; ragged shards (n_pairs % world != 0) pad to the largest shard so that it is
    still a single collective.  With `comm` (an RcclComm or a OneShotGather) the collective is the C ABI's (hdn_allgather_offsets on RCCL, or the
    direct-write hdn_gather_offsets_oneshot);
    otherwise torch.distributed's all_gather_into_tensor on `group` (backend "nccl" = RCCL on ROCm, "gloo" in the
    CPU tests).  A world of one returns x_local unless `always_collective` (tests: run RCCL on a one-GPU box).
    """
    if x_local.dim() != 2:
        raise ValueError("x_local must be [B_local, D]")
    if comm is not None:
        world, rank = comm.world, comm.rank
    elif dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        return x_local
    if world == 1 and not always_collective:
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
    if comm is not None:
        out = comm.all_gather(buf)
    elif buf.is_cuda and dist.get_backend(group) == "gloo":
        out = buf.new_empty((world * cap, D))
        # gloo has no device all-gather: stage through the host (CPU tests / single-GPU dry runs only; on the GPUs the
        # backend is nccl = RCCL and the collective runs on device memory)
        host = out.cpu()
        dist.all_gather_into_tensor(host, buf.cpu(), group=group)
        out.copy_(host)
    else:
        out = buf.new_empty((world * cap, D))
        dist.all_gather_into_tensor(out, buf, group=group)
    if n_pairs == world * cap:
        return out
    return torch.cat([out[r * cap: r * cap + (e - s)] for r, (s, e) in enumerate(sizes)], dim=0)


def sharded_offsets(net, data: dict, group=None, comm=None, always_collective: bool = False) -> torch.Tensor:
    """Run the homography head on this rank's shard of `data` (dict of [B, ...] tensors, all ranks hold
    the same global batch) and return the gathered [B, 8] corner offsets."""
    from .homo_model import homo_stages

    if comm is not None:
        world, rank = comm.world, comm.rank
    else:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = data["org_imgs"].shape[0]
    s, e = shard_range(B, rank, world)
    local = {k: v[s:e].contiguous() for k, v in data.items() if k in ("org_imgs", "input_tensors", "h4p", "patch_indices")}
    x = homo_stages(net, local)["x"]
    return all_gather_offsets(x, B, group, comm, always_collective)
