#!/usr/bin/env python
"""Correctness (against fp64) and speed of the 2-way fp16 split GEMM (csrc/gemm_h2.hip) next to the bf16x3
and fp32-MFMA GEMMs at the C3 shapes.  python tools/h2_bench.py [--iters 10]"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from omnitokenizer_amd import _lib, ops  # noqa: E402
from tools.x3_bench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--clips", type=int, default=32)
    ap.add_argument("--tiles", default="1")
    ap.add_argument("--skip-check", action="store_true")
    ap.add_argument("--data", default="randn", choices=["randn", "zeros", "const"],
                    help="operand values of the speed section: the DVFS check of MI355X_MICROARCH.md (same binary, "
                         "zero-filled inputs toggle fewer bits -> higher clocks if the kernel is power-bound)")
    a = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)  # noqa: E731
    print("== correctness vs fp64 (max abs err) ==")
    for tile in (() if a.skip_check else (1, 3, 4)):
        _lib.set_option("h2_tile", tile)
        for (M, N, K, amp) in ((1000, 512, 512, 1.0), (4096, 1024, 512, 1.0), (777, 512, 1408, 1.0), (2048, 192, 512, 1.0),
                               (300, 64, 32, 1.0), (2000, 512, 512, 1e4), (2000, 512, 512, 1e-5)):
            x = r(M, K) * amp
            x[:, :4] *= 50.0   # outlier channels
            w = r(N, K) * 0.05
            w[:7] *= 30.0
            b = r(N)
            res = r(M, N)
            ref = x.double() @ w.double().T
            pk = ops.h2_pack_weight(w)
            bound = float(x.abs().max())
            e32 = (ops.linear(x, w).double() - ref).abs().max().item()
            ex3 = (ops.linear_x3(x, w).double() - ref).abs().max().item()
            eh2 = (ops.linear_h2(x, pk, bound).double() - ref).abs().max().item()
            eh2l = (ops.linear_h2(x, pk, bound * 1000.0).double() - ref).abs().max().item()   # loose bound
            ref2 = ref + b.double() + res.double()
            eh2b = (ops.linear_h2(x, pk, bound, bias=b, residual=res).double() - ref2).abs().max().item()
            print(f"tile {tile} M{M} N{N} K{K} amp {amp:g}: fp32-mfma {e32:.3e}  x3 {ex3:.3e}  h2 {eh2:.3e}  "
                  f"h2 loose-bound {eh2l:.3e}  h2+bias+res {eh2b:.3e}  scale {ref.abs().max().item():.3g}")
        M, K, inner = 1500, 512, 1365
        x = r(M, K)
        w1 = r(2 * inner, K) * 0.05
        wp = ops.pack_geglu_weight(w1, 1408)
        pk = ops.h2_pack_weight(wp)
        h = x.double() @ w1.double().T
        ref = torch.nn.functional.gelu(h[:, inner:]) * h[:, :inner]
        o32 = ops.linear_geglu(x, wp)
        oh2 = ops.linear_h2(x, pk, float(x.abs().max()), geglu=True)
        print(f"tile {tile} GEGLU: fp32-mfma {(o32[:, :inner].double() - ref).abs().max().item():.3e}  "
              f"h2 {(oh2[:, :inner].double() - ref).abs().max().item():.3e}  pad max {oh2[:, inner:].abs().max().item():.1e}")
        M, K = 2500, 512
        x = r(M, K) * 2 + 0.3
        gam, bet = r(K) * 0.2 + 1, r(K) * 0.1
        w = r(1536, K) * 0.05
        pk = ops.h2_pack_weight(w)
        st = ops.row_stats(x)
        y = ops.layernorm(x, gam, bet)
        ref = torch.cat([y.double() @ w[:512].double().T, x.double() @ w[512:].double().T], 1)
        lnb = math.sqrt(K) * float(gam.abs().max()) + float(bet.abs().max())
        o = ops.linear_h2(x, pk, float(x.abs().max()), ln=(st, gam, bet), ln_cols=512, ln_bound=lnb)
        print(f"tile {tile} fused LN q|kv: err vs fp64 {(o.double() - ref).abs().max().item():.3e}")
    _lib.set_option("h2_tile", 0)
    x = r(8192, 512)
    w = r(512, 512) * 0.05
    pk = ops.h2_pack_weight(w)
    big = ops.linear_h2(x, pk, 8.0)
    _lib.set_option("h2_tile", 4)
    small = ops.linear_h2(x[:100].contiguous(), pk, 8.0)
    _lib.set_option("h2_tile", 0)
    print("tiling independence (bitwise):", bool(torch.equal(big[:100], small)))

    if a.data != "randn":
        r0 = r
        if a.data == "zeros":
            r = lambda *s: torch.zeros(*s, device="cuda")  # noqa: E731
        else:
            r = lambda *s: torch.full(s, 0.5, device="cuda")  # noqa: E731
    L = a.clips * 5120
    D = 512
    x = r(L, D)
    h = r(L, 1408)
    x2 = r(L, D)
    gam = r(D)
    st = ops.row_stats(x)
    wff = ops.pack_geglu_weight(r(2730, D) * 0.04, 1408)
    wfo = r(D, 1408) * 0.04
    wq = r(D, D) * 0.04
    wkv = r(2 * D, D) * 0.04
    wqkv = r(3 * D, D) * 0.04
    pff, pfo, pq, pkv, pqkv = (ops.h2_pack_weight(t) for t in (wff, wfo, wq, wkv, wqkv))
    lnb = math.sqrt(D) * float(gam.abs().max())
    shapes = {
        "ff_in": (lambda: ops.linear_x3(x, wff, geglu=True), lambda: ops.linear_h2(x, pff, 8.0, geglu=True), 2.0 * L * D * 2730),
        "ff_in_ln": (lambda: ops.linear_x3(x, wff, geglu=True, ln=(st, gam, None)),
                     lambda: ops.linear_h2(x, pff, 8.0, geglu=True, ln=(st, gam, None), ln_bound=lnb), 2.0 * L * D * 2730),
        "ff_out": (lambda: ops.linear_x3(h, wfo, residual=x2), lambda: ops.linear_h2(h, pfo, 8.0, residual=x2), 2.0 * L * D * 1365),
        "q": (lambda: ops.linear_x3(x, wq), lambda: ops.linear_h2(x, pq, 8.0), 2.0 * L * D * D),
        "kv": (lambda: ops.linear_x3(x, wkv), lambda: ops.linear_h2(x, pkv, 8.0), 2.0 * L * D * 2 * D),
        "qkv_ln": (lambda: ops.linear_x3(x, wqkv, ln=(st, gam, None), ln_cols=512),
                   lambda: ops.linear_h2(x, pqkv, 8.0, ln=(st, gam, None), ln_cols=512, ln_bound=lnb), 2.0 * L * D * 3 * D),
        "out_res": (lambda: ops.linear_x3(x, wq, residual=x2), lambda: ops.linear_h2(x, pq, 8.0, residual=x2), 2.0 * L * D * D),
    }
    print("== speed at L =", L, "==")
    for name, (fx3, fh2, flops) in shapes.items():
        ms = timeit(fx3, a.iters)
        line = f"{name:9s} x3 {ms:.3f} ms {flops / ms / 1e9:.0f} TF |"
        for t in [int(v) for v in a.tiles.split(",")]:
            _lib.set_option("h2_tile", t)
            ms = timeit(fh2, a.iters)
            line += f" h2[t{t}] {ms:.3f} ms {flops / ms / 1e9:.0f} TF |"
        _lib.set_option("h2_tile", 0)
        print(line, flush=True)


if __name__ == "__main__":
    main()
