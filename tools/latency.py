#!/usr/bin/env python
"""Small-batch latency of encode+decode, eager launches vs one captured HIP graph replay (torch.cuda.CUDAGraph =
hipGraph on ROCm; asserts that the replay's ids and pixels equal the eager ones).  One 256^2 image: ~2.0 ms either way on
the MI355X -- the ~170 launches are queued faster than the GPU drains them, so these sizes are bound by kernel latency, not
by launch overhead.  Since r06 every call size runs the plane data flow (thin GEMM tiles for small calls); --pl-min-tokens N
routes calls below N tokens to the fp32-activation flow for A/B (profiles/r06_small_calls.txt).
    python tools/latency.py [--frames 1|17] [--batch 1] [--resolution 256]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from omnitokenizer_amd import OmniTokenizer_VQGAN, make_args, synth  # noqa: E402
from omnitokenizer_amd.config import OmniTokConfig  # noqa: E402


def wall(fn, iters):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--resolution", type=int, default=256)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--pl-min-tokens", type=int, default=None)
    a = ap.parse_args()
    if a.pl_min_tokens is not None:
        from omnitokenizer_amd import _lib
        _lib.set_option("pl_min_tokens", a.pl_min_tokens)
    args = make_args(2, resolution=a.resolution)
    cfg = OmniTokConfig.from_args(args)
    m = OmniTokenizer_VQGAN(args)
    m.load_state_dict(synth.synth_state_dict(cfg, 0), strict=True)
    m = m.cuda().eval()
    is_image = a.frames == 1
    x = (synth.synth_image(a.batch, a.resolution) if is_image
         else synth.synth_video(a.batch, a.frames, a.resolution)).cuda()

    def step():
        ids = m.encode(x, is_image)
        return ids, m.decode(ids, is_image)

    for _ in range(3):
        ids0, rec0 = step()
    eager = wall(step, a.iters)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        ids_g, rec_g = step()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(ids_g, ids0) and torch.equal(rec_g, rec0), "graph replay differs from eager"
    graph = wall(g.replay, a.iters)
    patches = ids0.numel()
    print(f"B={a.batch} frames={a.frames} {a.resolution}x{a.resolution}: eager {eager:.3f} ms, "
          f"hipGraph replay {graph:.3f} ms ({eager / graph:.2f}x), {patches / graph * 1e3:.0f} patches/s")


if __name__ == "__main__":
    main()
