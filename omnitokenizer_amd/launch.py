"""One-process-per-GPU launch protocol of the clip-sharded path (bench.py --gpus N, and any caller
that wants the same contract).

The reference's launch contract is environment based (ddp_utils.py:333-364: RANK / WORLD_SIZE /
LOCAL_RANK / MASTER_ADDR / MASTER_PORT set by the launcher, `init_process_group('nccl')`).  This
module keeps that contract and adds the piece a plain `python bench.py --gpus N` needs: when the
process was NOT started by a launcher (WORLD_SIZE unset) and N > 1 it re-executes itself through
`torch.distributed.run` on 127.0.0.1 with one rank per GPU.

The timed protocol (`timed_sharded_steps`) is the one the driver's contract asks for: W untimed
warm-up steps, barrier + device synchronise, exactly K steps, barrier + synchronise, MAX over ranks.
A step = encode(local shard) -> all-gather of the ids (the path's only collective; RCCL on the
GPU, gloo in the CPU tests) -> decode(local shard of the gathered ids).  After the timed region every
rank computes the CRC-32 of the gathered id tensor and the ranks compare them, so a collective
that silently returned garbage (or a world that is smaller than asked) fails the run.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
import zlib
from dataclasses import dataclass, field
from typing import Callable, List, Optional

import torch
import torch.distributed as dist

from . import dist as od


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launched_by_torchrun() -> bool:
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def respawn_command(script: str, argv: List[str], nproc: int, port: Optional[int] = None) -> List[str]:
    """The command `python script argv...` turns into when it has to start its own ranks."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script, *argv]


def maybe_respawn(script: str, argv: List[str], nproc: int) -> Optional[int]:
    """If `nproc` > 1 and no launcher started this process: run the ranks, return their exit code.
    Returns None when the caller should simply continue (single rank, or already a rank)."""
    if nproc <= 1 or launched_by_torchrun():
        return None
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "8")
    r = subprocess.run(respawn_command(script, argv, nproc), env=env)
    return r.returncode


@dataclass
class RankInfo:
    rank: int = 0
    world: int = 1
    local_rank: int = 0
    backend: str = "none"
    abandoned_thread: bool = False


def init_ranks(expected_world: int, backend: str = "nccl", set_cuda_device: bool = True) -> RankInfo:
    """Reads the launcher's environment, binds this rank to its GPU and creates the process group.
    Raises if the world the launcher created is not the one the command line asked for."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != expected_world:
        raise RuntimeError(f"asked for {expected_world} ranks but the launcher created WORLD_SIZE={world}")
    if set_cuda_device:
        torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend)   # "nccl" is RCCL on ROCm
        if dist.get_world_size() != world:
            raise RuntimeError(f"process group has {dist.get_world_size()} ranks, expected {world}")
    return RankInfo(rank, world, local_rank, backend if world > 1 else "none")


class rank_errors:
    """`with rank_errors(info): ...` -- a rank that fails says which rank it is and why on stderr BEFORE the launcher
    tears the job down (torchrun's own report is "rank 1 exitcode 1" with no reason), then exits non-zero without
    waiting on collectives the other ranks will never complete."""

    def __init__(self, info: "RankInfo" = None):
        self.info = info

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is None or issubclass(et, SystemExit):
            return False
        import traceback
        rank = self.info.rank if self.info is not None else int(os.environ.get("RANK", "0"))
        world = self.info.world if self.info is not None else int(os.environ.get("WORLD_SIZE", "1"))
        sys.stderr.write(f"rank {rank} of {world}: {et.__name__}: {ev}\n")
        sys.stderr.write("".join(traceback.format_exception(et, ev, tb)))
        sys.stderr.flush()
        sys.stdout.flush()  # os._exit skips the interpreter's own flush: partial JSON lines would be lost
        if world > 1:
            os._exit(1)   # do not run atexit / destructors that wait for the other ranks
        return False


@dataclass
class ShardedResult:
    seconds: float                  # max over ranks of the timed region
    steps: int
    ids_local: torch.Tensor = None
    rec_local: torch.Tensor = None
    n_total: int = 0
    ids_crc: int = 0                # CRC-32 of the gathered ids (equal on every rank, checked)
    allgather_ms: Optional[float] = None   # one start() + wait() of the step's own persistent gather object
    per_rank_ms: Optional[List[float]] = None   # every rank's own ms per step of the timed region (rank order)
    gather_impl: str = "none"
    world_seen: int = 1
    extra: dict = field(default_factory=dict)


def _barrier(info: RankInfo, on_gpu: bool):
    if info.world > 1:
        if on_gpu:
            dist.barrier(device_ids=[info.local_rank])
        else:
            dist.barrier()
    if on_gpu:
        torch.cuda.synchronize()


def timed_sharded_steps(info: RankInfo, encode: Callable[[torch.Tensor], torch.Tensor],
                        decode: Callable[[torch.Tensor], torch.Tensor], x_local: torch.Tensor,
                        steps: int, warmup: int, native_gather: bool = False) -> ShardedResult:
    """Weak-scaling protocol: every rank owns `x_local` (its clips), n_total = world * local.
    native_gather: issue the id all-gather from C++ (dist.NativeIdGather: ncclAllGather through libomnitok.so on a side
    stream) instead of torch.distributed's all_gather_into_tensor."""
    on_gpu = x_local.is_cuda
    b_local = x_local.shape[0]
    n_total = b_local * info.world
    lo, hi = od.shard_range(n_total, info.rank, info.world)
    state = {"gather": None}
    trace = [] if os.environ.get("OMNITOK_TRACE_STEP") else None  # tests: order of the phases of the last step

    def step():
        if trace is not None:
            trace.clear()
        ids_local = encode(x_local)
        if trace is not None:
            trace.append("encode")
        g = None
        if info.world > 1:
            # the one collective of the path, issued asynchronously: decode consumes only the local shard, so the
            # gather overlaps it and is waited for only when the gathered tensor is needed (its CRC, the caller)
            if state["gather"] is None:
                if native_gather:
                    state["comm"] = od.NativeComm(device=ids_local.device)
                    state["gather"] = od.NativeIdGather(n_total, ids_local.shape[1:], ids_local.device, state["comm"])
                else:
                    state["gather"] = od.IdGather(n_total, ids_local.shape[1:], ids_local.device)
            g = state["gather"].start(ids_local)
            if trace is not None:
                trace.append("gather_start")
        rec = decode(ids_local)
        if trace is not None:
            trace.append("decode")
        if g is not None:
            state["ids_all"] = g.wait()
            if trace is not None:
                trace.append("gather_wait")
        else:
            state["ids_all"] = ids_local
        return ids_local, rec

    for _ in range(warmup):
        step()
    _barrier(info, on_gpu)
    t0 = time.perf_counter()
    for _ in range(steps):
        ids, rec = step()
    if on_gpu:
        torch.cuda.synchronize()
    own = time.perf_counter() - t0     # this rank alone (a straggler shows here, not in the barrier-closed time)
    _barrier(info, on_gpu)
    dt = time.perf_counter() - t0
    res = ShardedResult(seconds=dt, steps=steps, ids_local=ids, rec_local=rec, n_total=n_total,
                        world_seen=info.world, per_rank_ms=[round(own / steps * 1e3, 3)])
    if trace is not None:
        res.extra["step_trace"] = list(trace)
    ids_all = state["ids_all"]
    res.ids_crc = zlib.crc32(ids_all.cpu().numpy().tobytes())
    if info.world > 1:
        dev = x_local.device
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        res.seconds = float(tt.item())
        # (1) the gathered tensor holds this rank's own ids in its shard
        if not torch.equal(ids_all[lo:hi], ids):
            raise RuntimeError(f"rank {info.rank}: gathered ids differ from the locally encoded ids")
        # (2) every rank holds the same gathered tensor; (3) the group really has `world` members
        crcs = torch.zeros(info.world, dtype=torch.int64, device=dev)
        mine = torch.tensor([res.ids_crc], dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(crcs, mine)
        crcs = crcs.cpu().tolist()
        if len(set(crcs)) != 1:
            raise RuntimeError(f"gathered-id CRCs differ across ranks: {crcs}")
        ones = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(ones)
        res.world_seen = int(ones.item())
        if res.world_seen != info.world:
            raise RuntimeError(f"collective saw {res.world_seen} ranks, expected {info.world}")
        owns = torch.zeros(info.world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(owns, torch.tensor([own / steps * 1e3], dtype=torch.float64, device=dev))
        res.per_rank_ms = [round(v, 3) for v in owns.cpu().tolist()]
        res.gather_impl = type(state["gather"]).__name__
        # cost of the collective alone (untimed region): the step's own persistent gather object, start() + wait() --
        # no allocation, no clone; what a step would pay if nothing overlapped it
        g, reps = state["gather"], 10
        g.start(ids).wait()
        _barrier(info, on_gpu)
        t1 = time.perf_counter()
        for _ in range(reps):
            g.start(ids).wait()
        if on_gpu:
            torch.cuda.synchronize()
        ag = torch.tensor([(time.perf_counter() - t1) / reps * 1e3], device=dev, dtype=torch.float64)
        dist.all_reduce(ag, op=dist.ReduceOp.MAX)
        res.allgather_ms = float(ag.item())
    return res


def probe_native_gather(info: RankInfo, ids_local: torch.Tensor, res: ShardedResult, timeout_s: float = 45.0):
    """The C++ ncclAllGather path (include/omnitok_comm.h, dist.NativeIdGather) exercised next to whichever gather the
    timed region used: communicator bootstrap through the existing process group, one gather compared (CRC-32) with
    the timed region's gathered ids, then its start() + wait() time.  Never fatal and never unbounded: it runs on a
    helper thread that is abandoned after `timeout_s` (the caller then leaves with os._exit once its line is out)."""
    import threading
    out = {"ok": False}

    def body():
        try:
            torch.cuda.set_device(ids_local.device)
            comm = od.NativeComm(device=ids_local.device)
            g = od.NativeIdGather(res.n_total, ids_local.shape[1:], ids_local.device, comm)
            got = g.start(ids_local).wait()
            torch.cuda.synchronize()
            crc = zlib.crc32(got.cpu().numpy().tobytes())
            reps = 10
            t = time.perf_counter()
            for _ in range(reps):
                g.start(ids_local).wait()
            torch.cuda.synchronize()
            out.update(ok=bool(crc == res.ids_crc), ids_crc32=crc, ms=round((time.perf_counter() - t) / reps * 1e3, 4),
                       rccl=comm.where, world=comm.world)
            out["_comm"] = comm   # keep the communicator alive until the process ends (destroying it is collective)
        except Exception as e:  # noqa: BLE001
            out["error"] = repr(e)

    th = threading.Thread(target=body, daemon=True)
    th.start()
    th.join(timeout_s)
    hung = th.is_alive()
    if hung:
        out["error"] = f"timeout after {timeout_s:.0f} s"
    # Every rank must leave the same way: a rank whose probe hung cannot take part in the closing barrier /
    # destroy_process_group, and the others would wait for it until the watchdog fires.  Agreement goes through the
    # rendezvous store (host side, TCP) -- not through a collective, which may be what is stuck.
    hung_ranks = _agree_on_hang(info, hung)
    if hung_ranks:
        info.abandoned_thread = True
        out["hung_ranks"] = hung_ranks
        if not hung:
            out["error"] = f"native gather probe hung on rank(s) {hung_ranks}"
            out["ok"] = False
    res.extra["native_comm"] = out.pop("_comm", None)
    return out


def _agree_on_hang(info: RankInfo, hung: bool, timeout_s: float = 30.0):
    """-> sorted list of ranks whose probe thread hung (every rank gets the same list; a rank that never reports counts as
    hung).  world 1: [0] or []."""
    if info.world <= 1 or not dist.is_initialized():
        return [info.rank] if hung else []
    import datetime
    try:
        store = dist.distributed_c10d._get_default_store()
        # one key set per call: every rank calls this the same number of times, so the sequence number agrees across ranks and a
        # second probe in the same process group never reads the previous call's values
        seq = getattr(info, "_probe_seq", 0)
        info._probe_seq = seq + 1
        store.set(f"omnitok/native_probe/{seq}/{info.rank}", "1" if hung else "0")
        bad = []
        for r in range(info.world):
            key = f"omnitok/native_probe/{seq}/{r}"
            try:
                store.wait([key], datetime.timedelta(seconds=timeout_s))
                if store.get(key) != b"0":
                    bad.append(r)
            except Exception:  # noqa: BLE001  (the rank never reported)
                bad.append(r)
        return bad
    except Exception:  # noqa: BLE001  (no store: fall back to this rank's own view)
        return [info.rank] if hung else []


def finish(info: RankInfo, on_gpu: bool = True, native_timed: bool = False):
    """Orderly teardown; when a native-gather probe thread hangs inside a collective (on ANY rank, see _agree_on_hang) every
    rank leaves through os._exit -- status 0 when the timed region used torch.distributed's gather (the line is valid, the
    probe's `error` field says what happened), status 3 when the timed region itself was the native gather."""
    if getattr(info, "abandoned_thread", False):
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(3 if native_timed else 0)
    if info.world > 1:
        _barrier(info, on_gpu)
        dist.destroy_process_group()
