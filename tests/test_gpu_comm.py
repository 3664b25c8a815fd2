"""GPU (-m gpu): the C++ RCCL gather of include/omnitok_comm.h on the one GPU a test box has -- a communicator of world
size 1 exercises the whole plumbing (dlopen of librccl.so, ncclGetUniqueId, ncclCommInitRank, the narrowing kernel,
ncclAllGather on a side stream, the widening kernel, stream ordering against the producer).  The N > 1 semantics of the
gather protocol are covered on CPU by the gloo tests (tests/test_dist_gloo.py) and on hardware by `bench.py --gpus N`,
which checks this path against torch.distributed's result (native_gather_probe)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_native_comm_world1_allgather_ids():
    from omnitokenizer_amd import _lib
    from omnitokenizer_amd import dist as od
    comm = od.NativeComm()
    assert comm.world == 1 and comm.rank == 0 and "rccl" in comm.where
    lib = _lib.load()
    assert lib.omnitok_comm_world(comm.handle) == 1 and lib.omnitok_comm_rank(comm.handle) == 0
    g = od.NativeIdGather(32, (5, 32, 32), torch.device("cuda"), comm)
    for rep in range(3):   # the staging block is allocated once and reused
        ids = torch.randint(0, 8192, (32, 5, 32, 32), device="cuda", dtype=torch.int64)
        ids = ids * 1 + 0   # produced by a kernel still in flight on the current stream when start() is called
        out = g.start(ids).wait()
        torch.cuda.synchronize()
        assert out.dtype == torch.int64 and torch.equal(out, ids)
    # raw i32 entry point, in place (send == recv + rank * count, as NCCL allows)
    buf = torch.arange(1000, dtype=torch.int32, device="cuda")
    _lib.check(lib.omnitok_comm_allgather_i32(comm.handle, ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(buf.data_ptr()),
                                              1000, torch.cuda.current_stream().cuda_stream), "allgather_i32")
    torch.cuda.synchronize()
    assert torch.equal(buf.cpu(), torch.arange(1000, dtype=torch.int32))
    with pytest.raises(ValueError):
        od.NativeIdGather(33, (4,), torch.device("cuda"), type("C", (), {"world": 2, "rank": 0})())
    comm.close()


def test_native_gather_in_the_sharded_step_protocol():
    """launch.timed_sharded_steps at world 1 never gathers; the probe used by bench.py at N > 1 is exercised directly."""
    from omnitokenizer_amd import launch
    info = launch.RankInfo()
    ids = torch.randint(0, 8192, (4, 5, 8, 8), device="cuda", dtype=torch.int64)
    import zlib
    res = launch.ShardedResult(seconds=1.0, steps=1, ids_local=ids, n_total=4, ids_crc=zlib.crc32(ids.cpu().numpy().tobytes()))
    out = launch.probe_native_gather(info, ids, res, timeout_s=60.0)
    assert out.get("ok") is True and out["world"] == 1 and out["ms"] > 0, out
