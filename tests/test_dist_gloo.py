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


def _gather_worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from omnitokenizer_amd import dist as od
        lo, hi = od.shard_range(n_total, rank, world)
        full = (torch.arange(n_total * 6, dtype=torch.int64).reshape(n_total, 2, 3) * 7) % 8192
        g = od.IdGather(n_total, (2, 3), torch.device("cpu"))
        ok = True
        for step in range(3):  # the buffers are reused every step
            mine = (full[lo:hi] + step) % 8192
            g.start(mine)
            ok = ok and g.work is not None            # asynchronous: nothing waited for yet
            out = g.wait()
            ok = ok and out.dtype == torch.int64 and torch.equal(out, (full + step) % 8192)
        ok = ok and g.even == (n_total % world == 0) and g.send.dtype == torch.int32 and g.recv.dtype == torch.int32
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 5, 1])
def test_id_gather_async_even_and_ragged(n_total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert dict(q.get(timeout=5) for _ in range(world)) == {0: True, 1: True}


def test_c4_shards_are_32_clips_each():
    # BASELINE config C4: 256 clips on 8 GPUs = the C3 load per GPU, even shards (no padding, no per-rank copies)
    assert [shard_range(256, r, 8) for r in range(8)] == [(32 * r, 32 * r + 32) for r in range(8)]


def test_sharded_job_pins_one_data_flow_for_all_ranks():
    """the plane / small-call switch ("pl_min_tokens") used to be decided per call from the LOCAL token count,
    so a ragged shard could round a clip differently than its neighbours.  dist.data_flow_for decides from the global
    batch; pin_data_flow applies it through the per-engine option and reads the process default through the C ABI."""
    from omnitokenizer_amd import _lib
    from omnitokenizer_amd.dist import data_flow_for, pin_data_flow
    assert _lib.get_option("pl_min_tokens") == 0   # r06: one data flow at every size is the default ...
    assert data_flow_for(9, 5120, 8, 0) == 0 and data_flow_for(1, 1024, 1, 0) == 0
    thr = 12288                                    # ... and the rule still holds for whoever sets a threshold (rounds 4-5: 12288)
    # C4: 256 clips of 5120 tokens on 8 ranks -> planes; 9 clips on 8 ranks: largest shard 2 clips = 10240 < 12288 -> the
    # small-call flow on EVERY rank (also the ranks that hold 1 clip); 17 clips on 8 ranks: largest 3 clips -> planes everywhere
    assert data_flow_for(256, 5120, 8, thr) == 0
    assert data_flow_for(9, 5120, 8, thr) == 1 << 30
    assert data_flow_for(17, 5120, 8, thr) == 0
    assert data_flow_for(12, 1024, 1, thr) == 0 and data_flow_for(11, 1024, 1, thr) == 1 << 30

    class FakeModel:
        def set_option(self, name, value):
            self.got = (name, value)
    m = FakeModel()
    assert pin_data_flow(m, 2, 5120) == 0 and m.got == ("pl_min_tokens", 0)
    try:
        _lib.set_option("pl_min_tokens", 12288)
        assert pin_data_flow(m, 2, 5120) == 1 << 30 and m.got == ("pl_min_tokens", 1 << 30)
    finally:
        _lib.set_option("pl_min_tokens", 0)
    with pytest.raises(ValueError):
        _lib.get_option("no_such_option")


def test_shard_range_partitions():
    for n in (0, 1, 7, 32, 256):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_self_spawning_launcher_gloo():
    """`python <script> --gpus 2` with no launcher around it starts its own two ranks (the command form
    the driver uses for bench.py), runs the timed sharded-step protocol and cross-checks the gathered ids."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMNITOK_TRACE_STEP"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "dist_driver.py"), "--gpus", "2"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["world_seen"] == 2 and out["n_total"] == 4
    assert out["allgather_ms"] is not None and out["ids_local_shape"][0] == 2
    # every rank's own step time is on the result (a straggler would show), and the gather is the persistent object of the step
    assert len(out["per_rank_ms"]) == 2 and all(v > 0 for v in out["per_rank_ms"]) and out["gather_impl"] == "IdGather"
    # the id all-gather is off the critical path: issued after encode, waited for only AFTER the local decode
    assert out["step_trace"] == ["encode", "gather_start", "decode", "gather_wait"]
    # single-rank run of the same driver: no respawn, no collective
    r1 = subprocess.run([sys.executable, os.path.join(root, "tests", "dist_driver.py"), "--gpus", "1"],
                        capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r1.returncode == 0, r1.stderr[-2000:]
    out1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])
    assert out1["n_gpus"] == 1 and out1["allgather_ms"] is None


def test_failing_rank_names_itself():
    """A rank that raises prints `rank r of N: <error>` before the launcher tears the job down (torchrun alone reports only
    'exitcode 1'), and the job exits non-zero instead of hanging in the other rank's collective."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "dist_driver.py"), "--gpus", "2", "--fail-rank", "1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode != 0
    assert "rank 1 of 2: RuntimeError: injected failure on this rank" in r.stderr, r.stderr[-2000:]


def test_launcher_rejects_wrong_world(monkeypatch):
    from omnitokenizer_amd import launch
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(RuntimeError, match="asked for 2 ranks"):
        launch.init_ranks(2, backend="gloo", set_cuda_device=False)
    cmd = launch.respawn_command("bench.py", ["--gpus", "4"], 4, port=29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-3:] == ["bench.py", "--gpus", "4"]
