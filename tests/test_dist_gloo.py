"""CPU, world_size 2 over gloo: the clip-sharding + id all-gather logic of omnitokenizer_amd/dist.py
(the N>1 path of bench.py), with the oracle standing in for the per-rank encoder."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from omnitokenizer_amd.dist import shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from omnitokenizer_amd import dist as od
        from oracle import omnitok_oracle as orc
        from tests.helpers import GoldenCase
        c = GoldenCase("s2_sdpa_r64_vid")
        torch.set_num_threads(2)
        reps = -(-n_total // c.x.shape[0])
        x = torch.cat([c.x] * reps)[:n_total]
        x = x + 0.01 * torch.arange(n_total).view(-1, 1, 1, 1, 1) / n_total  # make clips distinct
        with torch.no_grad():
            enc = lambda xs: orc.encode(c.sd, xs, False, c.cfg)  # noqa: E731
            ids_all = od.encode_sharded(enc, x)
            lo, hi = od.shard_range(n_total, rank, world)
            ids_mine = enc(x[lo:hi]) if hi > lo else None
            rec = od.decode_local(lambda i: orc.decode(c.sd, i, False, c.cfg), ids_all) if hi > lo else None
        ok = ids_all.shape[0] == n_total and ids_all.dtype == torch.int64
        if ids_mine is not None:
            ok = ok and torch.equal(ids_all[lo:hi], ids_mine) and rec.shape[0] == hi - lo
        # every rank must hold the same gathered tensor
        ref = ids_all.clone()
        dist.broadcast(ref, src=0)
        ok = ok and torch.equal(ref, ids_all)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [4, 3, 1])
def test_sharded_encode_allgather_gloo(n_total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(world))
    assert res == {0: True, 1: True}


def test_shard_range_partitions():
    for n in (0, 1, 7, 32, 256):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
