"""Test driver for omnitokenizer_amd/launch.py: the same self-spawn + timed sharded-step protocol
bench.py uses for --gpus N, on CPU over gloo with the oracle standing in for the per-rank HIP
encoder/decoder (test infrastructure: this file lives under tests/).

    python tests/dist_driver.py --gpus 2 [--steps 1 --warmup 0 --clips 2]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--clips", type=int, default=2, help="clips per rank")
    ap.add_argument("--fail-rank", type=int, default=-1, help="this rank raises before the timed region (error-report test)")
    a = ap.parse_args()

    from omnitokenizer_amd import launch
    rc = launch.maybe_respawn(os.path.abspath(__file__), sys.argv[1:], a.gpus)
    if rc is not None:
        sys.exit(rc)
    with launch.rank_errors():
        run(a)


def run(a):
    from omnitokenizer_amd import launch
    info = launch.init_ranks(a.gpus, backend="gloo", set_cuda_device=False)
    if info.rank == a.fail_rank:
        raise RuntimeError("injected failure on this rank (tests/test_dist_gloo.py)")

    from oracle import omnitok_oracle as orc
    from tests.helpers import GoldenCase
    c = GoldenCase("s2_sdpa_r64_vid")
    torch.set_num_threads(2)
    reps = -(-a.clips // c.x.shape[0])
    x = torch.cat([c.x] * reps)[: a.clips].clone()
    # distinct clips per rank and per slot
    x += 0.01 * (info.rank * a.clips + torch.arange(a.clips)).view(-1, 1, 1, 1, 1) / (a.clips * max(info.world, 1))
    with torch.no_grad():
        res = launch.timed_sharded_steps(info, lambda xs: orc.encode(c.sd, xs, False, c.cfg),
                                         lambda i: orc.decode(c.sd, i, False, c.cfg), x, a.steps, a.warmup)
    if info.rank == 0:
        print(json.dumps({"n_gpus": info.world, "world_seen": res.world_seen, "ids_crc32": res.ids_crc,
                          "n_total": res.n_total, "allgather_ms": res.allgather_ms, "seconds": res.seconds,
                          "per_rank_ms": res.per_rank_ms, "gather_impl": res.gather_impl,
                          "ids_local_shape": list(res.ids_local.shape), "rec_local_shape": list(res.rec_local.shape),
                          "step_trace": res.extra.get("step_trace")}),
              flush=True)
    launch.finish(info, on_gpu=False)


if __name__ == "__main__":
    main()
