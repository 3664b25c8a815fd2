#!/usr/bin/env python
"""gemm_pp (both operands pre-split, LDS-DMA staging) vs gemm_h2 on the C3 GEMM shapes: the experiment behind
DESIGN.md (h)-1.  python tools/pp_bench.py [--data randn|const]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from omnitokenizer_amd import ops  # noqa: E402
from tools.x3_bench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--data", default="randn", choices=["randn", "const"])
    a = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(0)
    if a.data == "randn":
        r = lambda *s: torch.randn(*s, device="cuda", generator=g)  # noqa: E731
    else:
        r = lambda *s: torch.full(s, 0.5, device="cuda")  # noqa: E731
    # correctness on a small problem
    x, w = torch.randn(512, 512, device="cuda", generator=g), torch.randn(256, 512, device="cuda", generator=g) * 0.05
    ref = x.double() @ w.double().T
    out = ops.linear_pp(ops.h2_pack_weight(x), ops.h2_pack_weight(w), 512, 256)
    e_pp = (out.double() - ref).abs().max().item()
    e_h2 = (ops.linear_h2(x, ops.h2_pack_weight(w), float(x.abs().max())).double() - ref).abs().max().item()
    print(f"err vs fp64: pp {e_pp:.3e}  h2 {e_h2:.3e}")
    L = 32 * 5120
    for name, N, K in (("q/out", 512, 512), ("kv", 1024, 512), ("qkv", 1536, 512), ("ff_out", 512, 1408), ("ff_in", 2816, 512)):
        x = r(L, K)
        w = r(N, K) * 0.04
        pa, pw = ops.h2_pack_weight(x), ops.h2_pack_weight(w)
        fl = 2.0 * L * N * K
        ms_pp = timeit(lambda: ops.linear_pp(pa, pw, L, N), a.iters)
        ms_h2 = timeit(lambda: ops.linear_h2(x, pw, 8.0), a.iters)
        print(f"{name:7s} N{N} K{K}: pp {ms_pp:.3f} ms {fl / ms_pp / 1e9:.0f} TF | h2 {ms_h2:.3f} ms {fl / ms_h2 / 1e9:.0f} TF", flush=True)


if __name__ == "__main__":
    main()
