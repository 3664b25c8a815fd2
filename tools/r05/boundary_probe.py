"""GPU box: what does one dependent kernel boundary cost here?  A chain of n trivial kernels (torch elementwise add on 1 / 65536 /
16 M floats) eager and as a replayed torch.cuda.graph; per-kernel time from host-synchronised wall clock over many replays."""
import time

import torch

dev = "cuda"
for numel in (1, 65536, 1 << 24):
    x = torch.zeros(numel, device=dev)
    n = 200
    def chain():
        for _ in range(n):
            x.add_(1.0)
    chain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        chain()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / (20 * n) * 1e6
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        chain()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            chain()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / (50 * n) * 1e6
    print(f"numel {numel}: eager {eager:.2f} us per kernel, graph replay {graph:.2f} us per kernel "
          f"(bytes per kernel {numel * 8}: {numel * 8 / 6.3e6:.2f} us at 6.3 TB/s)", flush=True)
