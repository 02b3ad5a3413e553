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

    def comm_count(self) -> int:
        """The number of ranks RCCL itself reports for this communicator (ncclCommCount through hdn_rccl_comm_count)."""
        if self._h is None:
            raise _lib.HdnHipError("communicator destroyed")
        n = ctypes.c_int(0)
        _lib.check(_lib.load().hdn_rccl_comm_count(self._h, ctypes.byref(n)), "hdn_rccl_comm_count")
        return int(n.value)

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
    sharded_offsets (`comm=`).  Needs every pair of ranks to have peer access (one xGMI node): from_process_group checks
    hipDeviceCanAccessPeer for every pair first and hands back an RcclComm (with a warning) when any pair cannot.

    Failure handling: a call that waits 2 s for a peer in vain fills that peer's rows with NaN and poisons the object — status()
    is non-zero from then on and every later all_gather raises (the two-slot protocol cannot survive a wait that gave up).  Read
    check_status() at your next synchronisation point; all_gather does it for the previous call."""

    def __init__(self, world: int, rank: int, max_rows: int, exchange, device=None, group=None):
        self.world, self.rank, self.max_rows = int(world), int(rank), int(max_rows)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._group = group
        self._h = None
        h = ctypes.c_void_p()
        lib = _lib.load()
        with _lib.device_guard(self.device):
            # exchange() is a collective of the caller's: it runs on EVERY rank exactly once, whatever happened before it on this
            # one (a rank whose window could not be created hands out 64 zero bytes and raises afterwards), so that the ranks'
            # collective sequences stay identical.
            err, mine = None, ctypes.create_string_buffer(64)
            try:
                _lib.check(lib.hdn_gather_create(ctypes.byref(h), self.world, self.rank, self.max_rows * 32), "hdn_gather_create")
                _lib.check(lib.hdn_gather_handle(h, mine), "hdn_gather_handle")
            except Exception as e:
                err = e
            try:
                handles = exchange(mine.raw)
                if err is not None:
                    raise err
                if len(handles) != self.world or any(len(x) != 64 for x in handles):
                    raise ValueError("exchange() must return one 64-byte handle per rank")
                if any(x == b"\0" * 64 for x in handles):
                    raise _lib.HdnHipError("a peer could not create its gather window")
                _lib.check(lib.hdn_gather_connect(h, ctypes.c_char_p(b"".join(handles))), "hdn_gather_connect")
            except Exception:
                if h:                              # (the window is freed; peers that already mapped it must not use it)
                    lib.hdn_gather_destroy(h)
                raise
            self._h = h

    @staticmethod
    def peers_reachable(device, group=None):
        """(every pair of ranks can reach each other's device memory?, reason).  Each rank publishes (host name, PCI bus id of its
        device); a peer on another host, a peer device this process cannot see (HIP_VISIBLE_DEVICES isolation: hipIpc would still
        work, but nothing can be checked up front) or hipDeviceCanAccessPeer == 0 all count as unreachable.  Two ranks on ONE
        device (the one-GPU test rig) are reachable.  Collective over `group`."""
        import socket

        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        lib = _lib.load()
        buf = ctypes.create_string_buffer(64)
        _lib.check(lib.hdn_device_pci_bus_id(idx, buf, 64), "hdn_device_pci_bus_id")
        me = (socket.gethostname(), buf.value.decode())
        world = dist.get_world_size(group)
        box = [None] * world
        dist.all_gather_object(box, me, group=group)
        ok, why = True, ""
        for r, (host, pci) in enumerate(box):
            if r == dist.get_rank(group):
                continue
            if host != me[0]:
                ok, why = False, f"rank {r} runs on another host ({host})"
                break
            rc = lib.hdn_gather_peer_access(idx, pci.encode())
            if rc != 1:
                ok, why = False, (f"device {pci} of rank {r} is not visible to this process" if rc == -2 else
                                  f"hipDeviceCanAccessPeer({me[1]} -> {pci}) = {rc}")
                break
        verdicts = [None] * world
        dist.all_gather_object(verdicts, (ok, why), group=group)
        bad = [f"rank {r}: {w}" for r, (o, w) in enumerate(verdicts) if not o]
        return (not bad), "; ".join(bad)

    @classmethod
    def from_process_group(cls, max_rows: int, device=None, group=None, fallback: bool = True):
        """Collective.  Returns a OneShotGather, or — when some pair of ranks has no peer access, or a rank fails to map a peer's
        window — an RcclComm on the same group (fallback=True, with a warning; fallback=False raises)."""
        if not (dist.is_available() and dist.is_initialized()):
            return cls(1, 0, max_rows, (lambda mine: [mine]), device)
        world, rank = dist.get_world_size(group), dist.get_rank(group)

        def exchange(mine):
            box = [None] * world
            dist.all_gather_object(box, mine, group=group)
            return box

        def give_up(reason):
            msg = f"hdn_amd: one-shot gather unavailable ({reason})"
            if not fallback:
                raise _lib.HdnHipError(msg)
            import warnings
            warnings.warn(msg + "; using RCCL (hdn_allgather_offsets) instead")
            return RcclComm.from_process_group(device, group)

        if world > 1:
            ok, why = cls.peers_reachable(device, group)
            if not ok:
                return give_up(why)
        obj, err = None, ""
        try:
            obj = cls(world, rank, max_rows, exchange, device, group)
        except Exception as e:       # (e.g. hipIpcOpenMemHandle refused): every rank must learn of it before anyone launches
            err = f"rank {rank}: {type(e).__name__}: {e}"
        if world > 1:
            box = [None] * world
            dist.all_gather_object(box, err, group=group)
            bad = [b for b in box if b]
            if bad:
                # Ranks that built their object and ranks that did not must keep issuing the SAME collectives: nothing has been
                # launched yet, so the survivors tear down locally (no barrier) and every rank goes on to the fallback together.
                if obj is not None:
                    obj.destroy(collective=False)
                return give_up("; ".join(bad))
        elif err:
            raise _lib.HdnHipError(err)
        return obj

    def all_gather(self, local: torch.Tensor) -> torch.Tensor:
        """local [Bl, 8] on this object's device -> [world * Bl, 8]; asynchronous on torch's current stream."""
        dev = _lib.require_device(local)
        if dev != self.device:
            raise _lib.HdnHipError(f"gather window bound to {self.device}, tensor on {dev}")
        if local.dim() != 2 or local.shape[1] != 8 or local.shape[0] == 0 or local.shape[0] > self.max_rows:
            raise ValueError(f"expected [1..{self.max_rows}, 8] corner offsets, got {tuple(local.shape)}")
        if self._h is None:
            raise _lib.HdnHipError("gather window destroyed")
        self.check_status()            # (of the calls before this one: a host read of pinned memory, no synchronisation)
        loc = local.detach().to(torch.float32).contiguous()
        out = torch.empty((self.world * loc.shape[0], 8), dtype=torch.float32, device=dev)
        with _lib.device_guard(dev):
            rc = _lib.load().hdn_gather_offsets_oneshot(self._h, _lib.ptr(loc), _lib.ptr(out), loc.shape[0], _lib.stream_ptr(dev))
        _lib.check(rc, "hdn_gather_offsets_oneshot")
        return out

    def status(self) -> int:
        """0, or non-zero (sticky) once a call gave up waiting for a peer (read it after the stream has drained)."""
        return int(_lib.load().hdn_gather_status(self._h)) if self._h is not None else 0

    def check_status(self):
        """Raise if any call so far timed out: the rows of the missing peer in that call's result are NaN and this object must
        not be used again."""
        st = self.status()
        if st:
            raise _lib.HdnHipError(f"one-shot gather: a peer did not deliver within 2 s (status {st}); the affected rows are NaN and "
                                   "this communicator is unusable — destroy it and rebuild, or fall back to RcclComm")

    def destroy(self, collective: bool = True, timeout_s: float = None):
        """Collective when the object came from a process group: every rank drains its device, then all meet at a barrier, and only
        then are the windows unmapped and freed (a peer's launch may still be storing into this rank's window before that).  The
        barrier is BOUNDED (timeout_s, default HDN_GATHER_DESTROY_TIMEOUT_S or 30 s): the usual reason to destroy a poisoned object
        is a peer that stopped answering, and a barrier with a dead peer never returns — after the timeout the teardown goes on
        locally with a warning, and the process group the barrier was issued on must not be used again (the orphaned barrier stays
        queued on it; any later collective there would be out of sequence with the peers).  collective=False: local teardown only, for objects no launch has used yet (from_process_group's
        failure path, where the ranks must keep issuing identical collectives)."""
        if self._h is not None:
            h, self._h = self._h, None
            torch.cuda.synchronize(self.device)
            if collective and (self._group is not None or (dist.is_available() and dist.is_initialized() and self.world > 1
                                                           and dist.get_world_size() == self.world)):
                import os
                import threading
                if timeout_s is None:
                    timeout_s = float(os.environ.get("HDN_GATHER_DESTROY_TIMEOUT_S", "30"))
                done = threading.Event()

                def meet():
                    try:
                        # a new thread's current device is 0: bind this rank's, or an RCCL barrier on a group that has not used a
                        # device yet would pick GPU 0 (and create a context there)
                        torch.cuda.set_device(self.device)
                        dist.barrier(group=self._group)
                    finally:
                        done.set()
                t = threading.Thread(target=meet, daemon=True)
                t.start()
                if not done.wait(timeout_s):
                    import warnings
                    warnings.warn(f"hdn_amd: OneShotGather.destroy: the peers did not reach the barrier within {timeout_s:g} s; "
                                  "freeing the gather window without them.  The barrier this rank issued is still queued on the process "
                                  "group: do NOT issue another collective on that group (it would pair with the peers out of sequence) — "
                                  "tear the group down, or rebuild it")
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
