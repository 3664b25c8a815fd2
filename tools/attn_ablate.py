#!/usr/bin/env python
"""Ablation arms of the hoisted-fragment attention kernel (csrc/attn_h2.hip, variant 3) at the C3 shape: wrong-result
measurement builds ("attn_h2_dbg", OMNITOK_ATTN_MEASUREMENT_BUILDS; library built by tools/build_meas_lib.sh).
  OMNITOK_LIB=tools/_bin/libomnitok_meas.so python tools/attn_ablate.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from omnitokenizer_amd import _lib, ops  # noqa: E402

ARMS = [(0, "full kernel"), (100, "full, launch bounds 4 workgroups / CU (128 registers, spills)"), (101, "full, bounds 2 (same code as 3)"),
        (102, "full, bounds 1 (188 registers: 2 workgroups / CU)"),
        (1, "no softmax arithmetic (P = S, split kept)"), (32, "no P split (lo = hi)"), (33, "no softmax, no split"),
        (2, "no S^T MFMAs"), (4, "no P.V MFMAs"), (6, "no MFMAs at all"), (7, "no MFMAs, no softmax"),
        (8, "no barrier / DMA wait"), (24, "no barrier, no DMA")]


def timeit(fn, iters=20, warm=3):
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


L, D = 32 * 5120, 512
g = torch.Generator(device="cuda").manual_seed(0)
q, kv = torch.randn(L, D, device="cuda", generator=g), torch.randn(L, 2 * D, device="cuda", generator=g)
qs = torch.ones(64, device="cuda")
cos, sin = (t.cuda() for t in ops.rope_table(1024))
packed, bounds = ops.attn_pack(q, kv[:, :D], kv[:, D:], 1024, 8, qs, qs, cos, sin, v_bound=8.0)
flops = 4.0 * (L // 1024) * 8 * 1024 * 1024 * 64
for var in (1, 3):
    _lib.set_option("attn_h2_variant", var)
    ms = timeit(lambda: ops.attn_spatial_h2(packed, bounds, L // 1024, 1024, 8))
    print(f"variant {var}: {ms:.4f} ms {flops / ms / 1e9:.1f} TF", flush=True)
_lib.set_option("attn_h2_variant", 3)
for rep in range(2):
    for dbg, what in ARMS:
        _lib.set_option("attn_h2_dbg", dbg)
        ms = timeit(lambda: ops.attn_spatial_h2(packed, bounds, L // 1024, 1024, 8))
        print(f"dbg {dbg:3d}: {ms:.4f} ms  ({flops / ms / 1e9:6.1f} TF-equivalent)  {what}", flush=True)
_lib.set_option("attn_h2_dbg", 0)
