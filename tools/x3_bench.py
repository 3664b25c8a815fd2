#!/usr/bin/env python
"""Correctness (against fp64) and speed of the in-kernel-split bf16x3 GEMM (csrc/gemm_x3.hip) next to the
fp32-MFMA GEMM, at the C3 shapes.  python tools/x3_bench.py [--iters 10] [--skip-check]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from omnitokenizer_amd import _lib, ops  # noqa: E402


def timeit(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--clips", type=int, default=32)
    ap.add_argument("--skip-check", action="store_true")
    ap.add_argument("--tiles", default="1,2,3")
    a = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)  # noqa: E731

    if not a.skip_check:
        print("== correctness vs fp64 (max abs err; scale = max |ref|) ==")
        for tile in sorted({int(t) for t in a.tiles.split(",")} | {4}):
            _lib.set_option("x3_tile", tile)
            for (M, N, K) in ((1000, 512, 512), (4096, 1024, 512), (777, 512, 1408), (2048, 192, 512),
                              (300, 64, 32)):
                x = r(M, K)
                w = r(N, K) * 0.05
                b = r(N)
                res = r(M, N)
                ref = x.double() @ w.double().T
                e32 = (ops.linear(x, w).double() - ref).abs().max().item()
                ex3 = (ops.linear_x3(x, w).double() - ref).abs().max().item()
                ref2 = ref + b.double() + res.double()
                ex3b = (ops.linear_x3(x, w, bias=b, residual=res).double() - ref2).abs().max().item()
                print(f"tile {tile} M{M} N{N} K{K}: fp32-mfma {e32:.3e}  x3 {ex3:.3e}  x3+bias+res {ex3b:.3e} "
                      f"scale {ref.abs().max().item():.2f}")
            # GEGLU
            M, K, inner = 1500, 512, 1365
            x = r(M, K)
            w1 = r(2 * inner, K) * 0.05
            wp = ops.pack_geglu_weight(w1, 1408)
            h = x.double() @ w1.double().T
            ref = torch.nn.functional.gelu(h[:, inner:]) * h[:, :inner]
            o32 = ops.linear_geglu(x, wp)
            ox3 = ops.linear_x3(x, wp, geglu=True)
            print(f"tile {tile} GEGLU: fp32-mfma {(o32[:, :inner].double() - ref).abs().max().item():.3e}  "
                  f"x3 {(ox3[:, :inner].double() - ref).abs().max().item():.3e}  pad max {ox3[:, inner:].abs().max().item():.1e}")
            # fused LN, q|kv split
            M, K = 2500, 512
            x = r(M, K) * 2 + 0.3
            gam, bet = r(K) * 0.2 + 1, r(K) * 0.1
            w = r(1536, K) * 0.05
            st = ops.row_stats(x)
            y = ops.layernorm(x, gam, bet)
            ref = torch.cat([y.double() @ w[:512].double().T, x.double() @ w[512:].double().T], 1)
            o = ops.linear_x3(x, w, ln=(st, gam, bet), ln_cols=512)
            o2 = torch.cat([ops.linear_x3(y, w[:512].contiguous()), ops.linear_x3(x, w[512:].contiguous())], 1)
            print(f"tile {tile} fused LN q|kv: err vs fp64 {(o.double() - ref).abs().max().item():.3e}; "
                  f"vs unfused x3 {(o - o2).abs().max().item():.3e}")
        _lib.set_option("x3_tile", 0)
        # batch independence: rows of a big problem == same rows computed alone with another tiling
        x = r(8192, 512)
        w = r(512, 512) * 0.05
        big = ops.linear_x3(x, w)
        _lib.set_option("x3_tile", 4)
        small = ops.linear_x3(x[:100].contiguous(), w)
        _lib.set_option("x3_tile", 0)
        print("tiling independence (bitwise):", bool(torch.equal(big[:100], small)))

    L = a.clips * 5120
    D = 512
    x = r(L, D)
    h = r(L, 1408)
    shapes = {
        "ff_in": (lambda: ops.linear_geglu(x, wff), lambda: ops.linear_x3(x, wff, geglu=True), 2.0 * L * D * 2730),
        "ff_out": (lambda: ops.linear(h, wfo, residual=x2), lambda: ops.linear_x3(h, wfo, residual=x2), 2.0 * L * D * 1365),
        "q": (lambda: ops.linear(x, wq), lambda: ops.linear_x3(x, wq), 2.0 * L * D * D),
        "kv": (lambda: ops.linear(x, wkv), lambda: ops.linear_x3(x, wkv), 2.0 * L * D * 2 * D),
        "qkv_ln": (None, lambda: ops.linear_x3(x, wqkv, ln=(st, gam, None), ln_cols=512), 2.0 * L * D * 3 * D),
        "out_res": (lambda: ops.linear(x, wq, residual=x2), lambda: ops.linear_x3(x, wq, residual=x2), 2.0 * L * D * D),
    }
    wff = ops.pack_geglu_weight(r(2730, D) * 0.04, 1408)
    wfo = r(D, 1408) * 0.04
    wq = r(D, D) * 0.04
    wkv = r(2 * D, D) * 0.04
    wqkv = r(3 * D, D) * 0.04
    x2 = r(L, D)
    gam = r(D)
    st = ops.row_stats(x)
    print("== speed at L =", L, "==")
    ms = timeit(lambda: ops.row_stats(x), a.iters)
    print(f"row_stats: {ms:.3f} ms  {L * D * 4 / ms / 1e6:.0f} GB/s")
    for name, (f32, fx3, flops) in shapes.items():
        line = f"{name:8s}"
        if f32 is not None:
            ms = timeit(f32, a.iters)
            line += f" fp32-mfma {ms:.3f} ms {flops / ms / 1e9:.0f} TF |"
        for tile in [int(t) for t in a.tiles.split(",")]:
            _lib.set_option("x3_tile", tile)
            ms = timeit(fx3, a.iters)
            line += f" x3[t{tile}] {ms:.3f} ms {flops / ms / 1e9:.0f} TF |"
        _lib.set_option("x3_tile", 0)
        print(line, flush=True)


if __name__ == "__main__":
    main()
