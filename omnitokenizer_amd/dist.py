"""Clip-sharded multi-GPU encode/decode: one process per GPU, no data-path collective except
the final all-gather of token ids (BASELINE.json north_star).

The reference's encode/decode performs no communication (every clip is independent end to end:
attention, PEG and VQ never cross the batch dimension), so the batch is partitioned into
contiguous shards, weights are replicated, and only the ids -- 4 bytes per token after narrowing
int64 -> int32 (n_codes <= 32768) -- are exchanged with one all_gather over RCCL/xGMI
(torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests).  Decoded pixels stay local.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of n items for `rank`; the first n % world ranks get one extra."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def data_flow_for(n_total: int, tokens_per_item: int, world: int, min_tokens: int) -> int:
    """The per-engine "pl_min_tokens" value that makes EVERY rank of a clip-sharded job take the data flow its largest
    shard would take on its own: 0 (plane flow everywhere) when ceil(n_total / world) items reach the threshold, else a
    value no call reaches (small-call flow everywhere).  The two flows round differently (include/omnitok.h
    "pl_min_tokens"), so without this a ragged shard -- or 2 clips per rank against 4 -- could give the same clip
    different latent bits on different ranks.  With the default threshold 0 (r06: one data flow at every size) this is
    always 0; it matters only for a process that sets the option."""
    largest = -(-int(n_total) // max(int(world), 1)) * int(tokens_per_item)
    return 0 if largest >= int(min_tokens) else (1 << 30)


def pin_data_flow(model, n_total: int, tokens_per_item: int, group=None) -> int:
    """Pins `model`'s engine (OmniTokenizer_VQGAN.set_option) to one data flow for a sharded job of n_total items of
    tokens_per_item tokens each, decided from the GLOBAL batch; returns the value set.  Call once before the step loop."""
    from . import _lib
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    value = data_flow_for(n_total, tokens_per_item, world, _lib.get_option("pl_min_tokens"))
    model.set_option("pl_min_tokens", value)
    return value


class IdGather:
    """The one collective of the path (reference launch contract ddp_utils.py:333-364: one process per GPU, a default
    process group): all-gather of the token ids, int32 on the wire, off the critical path.

        g = IdGather(n_total, tail_shape, device)      # buffers allocated once, reused every step
        g.start(ids_local)                             # one cast-copy into the send buffer, asynchronous all_gather
        ... decode(ids_local) ...                      # the local decode does not need the other ranks' ids
        ids_all = g.wait()                             # [n_total, ...] int64

    Even shards (n_total % world == 0, e.g. BASELINE config C4: 256 clips on 8 GPUs = 32 each) cost one cast-copy in,
    the collective and one cast-copy out; ragged shards copy each rank's valid rows out of the padded receive buffer."""

    def __init__(self, n_total: int, tail, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_total = int(n_total)
        self.tail = tuple(int(v) for v in tail)
        self.bmax = -(-self.n_total // self.world)
        self.even = self.n_total % self.world == 0
        self.send = torch.zeros((self.bmax,) + self.tail, dtype=torch.int32, device=device)
        self.recv = torch.empty((self.world * self.bmax,) + self.tail, dtype=torch.int32, device=device)
        self.out = torch.empty((self.n_total,) + self.tail, dtype=torch.int64, device=device)
        self.work = None

    def start(self, ids_local: torch.Tensor):
        assert self.work is None, "IdGather.start() called twice without wait()"
        lo, hi = shard_range(self.n_total, self.rank, self.world)
        assert ids_local.shape[0] == hi - lo and tuple(ids_local.shape[1:]) == self.tail, (ids_local.shape, lo, hi, self.tail)
        if hi > lo:
            self.send[: hi - lo].copy_(ids_local)  # int64 -> int32 in the copy kernel
        # asynchronous: NCCL/RCCL runs it on its own stream behind the copy; gloo on its worker thread
        self.work = dist.all_gather_into_tensor(self.recv, self.send, group=self.group, async_op=True)
        return self

    def wait(self) -> torch.Tensor:
        assert self.work is not None, "IdGather.wait() without start()"
        self.work.wait()
        self.work = None
        if self.even:
            self.out.copy_(self.recv)  # int32 -> int64
        else:
            for r in range(self.world):
                lo, hi = shard_range(self.n_total, r, self.world)
                if hi > lo:
                    self.out[lo:hi].copy_(self.recv[r * self.bmax: r * self.bmax + (hi - lo)])
        return self.out


class NativeComm:
    """An RCCL communicator owned by libomnitok.so (include/omnitok_comm.h): ncclGetUniqueId on rank 0, the 128 bytes
    broadcast through the EXISTING torch.distributed process group (whatever its backend), ncclCommInitRank on every
    rank's current device.  world == 1 needs no process group (single-GPU plumbing test)."""

    def __init__(self, group=None, device=None):
        import ctypes
        from . import _lib
        self._lib = _lib
        lib = _lib.load()
        where = ctypes.create_string_buffer(256)
        if not lib.omnitok_comm_available(where, 256):
            raise _lib.OmnitokError(f"native gather: {where.value.decode()}")
        self.where = where.value.decode()
        if dist.is_initialized():
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        else:
            self.rank, self.world = 0, 1
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        ident = (ctypes.c_ubyte * 128)()
        if self.rank == 0:
            _lib.check(lib.omnitok_comm_unique_id(ident), "comm_unique_id")
        if self.world > 1:
            t = torch.tensor(list(ident), dtype=torch.uint8)
            on_gpu = dist.get_backend(group) == "nccl"
            if on_gpu:
                t = t.to(self.device)
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast(t, src=src, group=group)
            ident = (ctypes.c_ubyte * 128)(*t.cpu().tolist())
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.omnitok_comm_create(ident, self.rank, self.world, ctypes.byref(h)), "comm_create")
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self._lib.load().omnitok_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class NativeIdGather:
    """IdGather with the collective issued from C++: `omnitok_comm_allgather_ids` (narrow int64 -> int32, ncclAllGather,
    widen) as three stream-ordered operations on a side stream that waits for the producer stream, so the host never
    blocks and the local decode overlaps the gather exactly as with IdGather.  Same start() / wait() surface; even
    shards only (BASELINE C4: 256 clips on 8 GPUs) -- ragged batches use IdGather."""

    def __init__(self, n_total: int, tail, device, comm: NativeComm):
        self.comm, self.world, self.rank = comm, comm.world, comm.rank
        self.n_total, self.tail = int(n_total), tuple(int(v) for v in tail)
        if self.n_total % self.world:
            raise ValueError("NativeIdGather needs n_total % world == 0 (use IdGather for ragged shards)")
        self.count = self.n_total // self.world
        for v in self.tail:
            self.count *= v
        self.out = torch.empty((self.n_total,) + self.tail, dtype=torch.int64, device=device)
        self.side = torch.cuda.Stream(device)
        self.pending = False
        self._keep = None

    def start(self, ids_local: torch.Tensor):
        assert not self.pending, "NativeIdGather.start() called twice without wait()"
        assert ids_local.dtype == torch.int64 and ids_local.is_contiguous() and ids_local.numel() == self.count
        lib = self.comm._lib.load()
        self.side.wait_stream(torch.cuda.current_stream(ids_local.device))
        ids_local.record_stream(self.side)
        self._keep = ids_local
        self.comm._lib.check(lib.omnitok_comm_allgather_ids(self.comm.handle, ids_local.data_ptr(), self.out.data_ptr(),
                                                            self.count, self.side.cuda_stream), "comm_allgather_ids")
        self.pending = True
        return self

    def wait(self) -> torch.Tensor:
        assert self.pending, "NativeIdGather.wait() without start()"
        torch.cuda.current_stream(self.out.device).wait_stream(self.side)
        self.pending, self._keep = False, None
        return self.out


def all_gather_ids(ids_local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """ids_local [b_local, ...] int64 -> [n_total, ...] int64 on every rank (blocking form of IdGather)."""
    world = dist.get_world_size(group)
    if world == 1:
        return ids_local
    return IdGather(n_total, ids_local.shape[1:], ids_local.device, group).start(ids_local).wait().clone()


def encode_sharded(encode_fn: Callable[[torch.Tensor], torch.Tensor], x_global_or_local: torch.Tensor,
                   n_total: Optional[int] = None, group=None, already_sharded: bool = False) -> torch.Tensor:
    """Runs encode_fn on this rank's shard of the batch and returns the ids of the WHOLE batch on
    every rank.  x is either the full batch (each rank slices its shard) or, with
    already_sharded=True, this rank's shard (n_total = global batch size)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if already_sharded:
        assert n_total is not None
        x_local = x_global_or_local
    else:
        n_total = x_global_or_local.shape[0]
        lo, hi = shard_range(n_total, rank, world)
        x_local = x_global_or_local[lo:hi]
    ids_local = encode_fn(x_local) if x_local.shape[0] > 0 else None
    if world == 1:
        return ids_local
    if ids_local is None:  # empty shard: learn the trailing shape from rank 0
        shp = torch.zeros(8, dtype=torch.int64, device=x_global_or_local.device)
        dist.broadcast(shp, src=0, group=group)
        nd = int(shp[0])
        ids_local = torch.zeros((0,) + tuple(int(v) for v in shp[1:1 + nd]), dtype=torch.int64,
                                device=x_global_or_local.device)
    elif n_total < world:  # someone has an empty shard: rank 0 publishes the shape
        shp = torch.zeros(8, dtype=torch.int64, device=ids_local.device)
        if rank == 0:
            shp[0] = ids_local.dim() - 1
            shp[1:ids_local.dim()] = torch.tensor(ids_local.shape[1:])
        dist.broadcast(shp, src=0, group=group)
    return all_gather_ids(ids_local, n_total, group)


def decode_local(decode_fn: Callable[[torch.Tensor], torch.Tensor], ids_all: torch.Tensor, group=None):
    """Decodes this rank's shard of the gathered ids; pixels are not gathered."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_range(ids_all.shape[0], rank, world)
    return decode_fn(ids_all[lo:hi])
