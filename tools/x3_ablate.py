#!/usr/bin/env python
"""Ablation timings of the 256x256 x3 GEMM (x3_dbg builds: wrong results, measurement only) and a plain
run for PMC passes.  python tools/x3_ablate.py [--pmc]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from omnitokenizer_amd import _lib, ops  # noqa: E402
from tools.x3_bench import timeit  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pmc", action="store_true", help="just run the kernels a few times (under rocprofv3 --pmc)")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--kind", default="x3", choices=["x3", "h2"])
a = ap.parse_args()
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)  # noqa: E731
L, D = 163840, 512
x = r(L, D)
wkv = r(2 * D, D) * 0.04
h = r(L, 1408)
wfo = r(D, 1408) * 0.04
_lib.set_option("x3_tile", 1)
_lib.set_option("h2_tile", 1)
if a.kind == "h2":
    pkv, pfo = ops.h2_pack_weight(wkv), ops.h2_pack_weight(wfo)
    names = {0: "full", 1: "no lo-plane arithmetic", 2: "no LDS stores (and no split)", 4: "no global loads",
             6: "no stores + no loads", 8: "no barrier", 14: "MFMA + fragment reads only"}
    for dbg, nm in names.items():
        _lib.set_option("h2_dbg", dbg)
        ms1 = timeit(lambda: ops.linear_h2(x, pkv, 8.0), a.iters)
        ms2 = timeit(lambda: ops.linear_h2(h, pfo, 8.0), a.iters)
        print(f"h2 dbg {dbg:2d} {nm:32s} kv(N1024,K512) {ms1:.3f} ms {2.0 * L * D * 2 * D / ms1 / 1e9:.0f} TF | "
              f"N512,K1408 {ms2:.3f} ms {2.0 * L * D * 1408 / ms2 / 1e9:.0f} TF", flush=True)
    _lib.set_option("h2_dbg", 0)
    sys.exit(0)
if a.pmc:
    for _ in range(3):
        ops.linear_x3(x, wkv)
        ops.linear_x3(h, wfo)
    torch.cuda.synchronize()
    sys.exit(0)
names = {0: "full", 1: "no split arithmetic", 2: "no LDS stores (and no split)", 4: "no global loads",
         6: "no stores + no loads", 8: "no barrier", 14: "MFMA + fragment reads only"}
for dbg, nm in names.items():
    _lib.set_option("x3_dbg", dbg)
    ms1 = timeit(lambda: ops.linear_x3(x, wkv), a.iters)
    ms2 = timeit(lambda: ops.linear_x3(h, wfo), a.iters)
    print(f"dbg {dbg:2d} {nm:32s} kv(N1024,K512) {ms1:.3f} ms {2.0 * L * D * 2 * D / ms1 / 1e9:.0f} TF | "
          f"N512,K1408 {ms2:.3f} ms {2.0 * L * D * 1408 / ms2 / 1e9:.0f} TF", flush=True)
# effective shader clock: clock64() span of workgroup 0..255 against the event-timed duration
import ctypes
trace = torch.zeros(512, dtype=torch.int64, device="cuda")
_lib.load().omnitok_debug_set_gemm_trace(ctypes.c_void_p(trace.data_ptr()))
for dbg in (16, 30):
    _lib.set_option("x3_dbg", dbg)
    for nm, fn in (("kv", lambda: ops.linear_x3(x, wkv)), ("K1408", lambda: ops.linear_x3(h, wfo))):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        t = trace.cpu().view(256, 2).double()
        cyc, wall = t[:, 0].mean().item(), t[:, 1].mean().item()
        print(f"dbg {dbg} {nm}: event {ms:.3f} ms; per-WG span {cyc / 1e6:.3f} M shader cycles, {wall / 1e3:.1f} k wall ticks "
              f"-> {cyc / wall * 100:.0f} MHz if the wall clock is 100 MHz; cycles / event time = {cyc / ms / 1e6:.2f} GHz")
_lib.load().omnitok_debug_set_gemm_trace(None)
_lib.set_option("x3_dbg", 0)
