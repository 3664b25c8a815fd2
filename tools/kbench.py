#!/usr/bin/env python
"""Single-kernel micro-benchmarks at the C3 shapes (B=32 clips, L=163840 tokens), through the same
C ABI the engine uses.  python tools/kbench.py [names...]  (default: all)
Prints one line per kernel: name ms TF/s-or-GB/s.  Used for A/B work on kernels and for rocprofv3
--pmc passes (run with --iters 3)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from omnitokenizer_amd import ops  # noqa: E402


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
    ap.add_argument("names", nargs="*")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--clips", type=int, default=32)
    ap.add_argument("--opt", action="append", default=[], help="name=value tuning knob (omnitok_set_option)")
    a = ap.parse_args()
    from omnitokenizer_amd import _lib
    for kv in a.opt:
        k, v = kv.split("=")
        _lib.set_option(k, int(v))
    L = a.clips * 5120
    D = 512
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)  # noqa: E731
    x = r(L, D)
    res = {}

    def want(n):
        return not a.names or n in a.names

    if want("gemm_ff_in"):
        w = ops.pack_geglu_weight(r(2730, D) * 0.04, 1408)
        ms = timeit(lambda: ops.linear_geglu(x, w), a.iters)
        res["gemm_ff_in"] = (ms, 2.0 * L * D * 2730 / ms / 1e9, "TF")
    if want("x3_ff_in"):
        w = ops.pack_geglu_weight(r(2730, D) * 0.04, 1408)
        ms = timeit(lambda: ops.linear_x3(x, w, geglu=True), a.iters)
        res["x3_ff_in"] = (ms, 2.0 * L * D * 2730 / ms / 1e9, "TF")
    if want("h2_ff_in"):
        w = ops.pack_geglu_weight(r(2730, D) * 0.04, 1408)
        pk = ops.h2_pack_weight(w)
        ms = timeit(lambda: ops.linear_h2(x, pk, 8.0, geglu=True), a.iters)
        res["h2_ff_in"] = (ms, 2.0 * L * D * 2730 / ms / 1e9, "TF")
    if want("h2_kv"):
        pk = ops.h2_pack_weight(r(2 * D, D) * 0.04)
        ms = timeit(lambda: ops.linear_h2(x, pk, 8.0), a.iters)
        res["h2_kv"] = (ms, 2.0 * L * D * 2 * D / ms / 1e9, "TF")
    if want("gemm_ff_out"):
        h = r(L, 1408)
        w = r(D, 1408) * 0.04
        ms = timeit(lambda: ops.linear(h, w, residual=x), a.iters)
        res["gemm_ff_out"] = (ms, 2.0 * L * D * 1365 / ms / 1e9, "TF")
    if want("gemm_q"):
        w = r(D, D) * 0.04
        ms = timeit(lambda: ops.linear(x, w), a.iters)
        res["gemm_q"] = (ms, 2.0 * L * D * D / ms / 1e9, "TF")
    if want("gemm_kv"):
        w = r(2 * D, D) * 0.04
        ms = timeit(lambda: ops.linear(x, w), a.iters)
        res["gemm_kv"] = (ms, 2.0 * L * D * 2 * D / ms / 1e9, "TF")
    if want("gemm_out"):
        w = r(D, D) * 0.04
        ms = timeit(lambda: ops.linear(x, w, residual=x), a.iters)
        res["gemm_out"] = (ms, 2.0 * L * D * D / ms / 1e9, "TF")
    if want("attn_spatial"):
        q = torch.nn.functional.normalize(r(L, 8, 64), dim=-1).reshape(L, D) * 8
        kv = r(L, 2 * D)
        kv[:, :D] = torch.nn.functional.normalize(kv[:, :D].reshape(L, 8, 64), dim=-1).reshape(L, D)
        ms = timeit(lambda: ops.attn_spatial(q, kv[:, :D], kv[:, D:], L // 1024, 1024, 8), a.iters)
        res["attn_spatial"] = (ms, 4.0 * (L // 1024) * 8 * 1024 * 1024 * 64 / ms / 1e9, "TF")
    if want("attn_h2"):
        q, kv = r(L, D), r(L, 2 * D)
        qs = torch.ones(64, device="cuda")
        cos, sin = (t.cuda() for t in ops.rope_table(1024))
        ms = timeit(lambda: ops.attn_pack(q, kv[:, :D], kv[:, D:], 1024, 8, qs, qs, cos, sin, v_bound=8.0), a.iters)
        res["attn_pack"] = (ms, 6.0 * L * D * 4 / ms / 1e6, "GB/s")
        packed, bounds = ops.attn_pack(q, kv[:, :D], kv[:, D:], 1024, 8, qs, qs, cos, sin, v_bound=8.0)
        for var in (1, 3, 5, 6, 7):
            _lib.set_option("attn_h2_variant", var)
            ms = timeit(lambda: ops.attn_spatial_h2(packed, bounds, L // 1024, 1024, 8), a.iters)
            res[f"attn_spatial_h2_v{var}"] = (ms, 4.0 * (L // 1024) * 8 * 1024 * 1024 * 64 / ms / 1e9, "TF")
        _lib.set_option("attn_h2_variant", 6)
    if want("attn_window"):
        qkv = r(L, 3 * D)
        bias = r(8, 64, 64)
        ms = timeit(lambda: ops.attn_window(qkv, bias, L // 1024, 32, 32, 8), a.iters)
        res["attn_window"] = (ms, 4.0 * L * 64 * D / ms / 1e9, "TF")
    if want("attn_temporal"):
        q, kv = r(L, D), r(L, 2 * D)
        qs = torch.ones(64, device="cuda")
        ms = timeit(lambda: ops.attn_temporal(q, kv[:, :D], kv[:, D:], L // 5, 5, 8, qs, qs, True), a.iters)
        res["attn_temporal"] = (ms, 4.0 * L * D * 4 / ms / 1e6, "GB/s")
    if want("peg3d"):
        w27, b = r(27, D) * 0.1, r(D) * 0.1
        ms = timeit(lambda: ops.peg3d(x, w27, b, (a.clips, 5, 32, 32), True), a.iters)
        res["peg3d"] = (ms, 2.0 * L * D * 4 / ms / 1e6, "GB/s")
    if want("layernorm"):
        gm = torch.ones(D, device="cuda")
        ms = timeit(lambda: ops.layernorm(x, gm, gm), a.iters)
        res["layernorm"] = (ms, 2.0 * L * D * 4 / ms / 1e6, "GB/s")
    if want("vq_argmin"):
        z = torch.nn.functional.normalize(r(L, 8), dim=-1)
        E = r(8192, 8)
        prep = ops.vq_prepare(E)
        ms = timeit(lambda: ops.vq_argmin(z, E, prep), a.iters)
        res["vq_argmin"] = (ms, 2.0 * L * 8192 * 8 / ms / 1e9, "TF")
    # ---- library baselines on the same shapes (only when asked for by name): PyTorch-ROCm fp32 ----
    # torch.mm dispatches to rocBLAS / hipBLASLt sgemm, F.scaled_dot_product_attention to its fp32 path
    if "lib_gemm" in a.names:
        torch.backends.cuda.matmul.allow_tf32 = False
        for nm, N, K in (("lib_sgemm_ff_in", 2730, D), ("lib_sgemm_kv", 2 * D, D), ("lib_sgemm_q", D, D),
                         ("lib_sgemm_ff_out", D, 1365)):
            aa, w = r(L, K), r(N, K) * 0.04
            ms = timeit(lambda: torch.mm(aa, w.t()), a.iters)
            res[nm] = (ms, 2.0 * L * N * K / ms / 1e9, "TF")
    if "lib_sdpa" in a.names:
        q4 = torch.nn.functional.normalize(r(a.clips * 5, 8, 1024, 64), dim=-1)
        ms = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q4, q4, q4, scale=8.0), max(2, a.iters // 3))
        res["lib_sdpa_spatial"] = (ms, 4.0 * a.clips * 5 * 8 * 1024 * 1024 * 64 / ms / 1e9, "TF")
    for k, (ms, rate, unit) in res.items():
        print(f"{k:16s} {ms:9.4f} ms  {rate:9.2f} {unit}")


if __name__ == "__main__":
    main()
