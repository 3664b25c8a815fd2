#!/usr/bin/env python
"""Sustained fp32-MFMA ceiling on this chip (v_mfma_f32_32x32x2_f32 only, random or zero operands,
1 or 2 waves per SIMD).  Context for roofline fractions: peak 157.3 TF assumes 2.4 GHz."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from omnitokenizer_amd import _lib  # noqa: E402

lib = _lib.load()
for fill in ("randn", "zeros"):
    x = torch.randn(4096, device="cuda") if fill == "randn" else torch.zeros(4096, device="cuda")
    for blocks, lds, label in ((256, 100 * 1024, "1 wave/SIMD"), (512, 60 * 1024, "2 waves/SIMD"),
                               (1024, 30 * 1024, "4 waves/SIMD")):
        out = torch.empty(blocks * 256, device="cuda")
        iters = 4000
        s = torch.cuda.current_stream().cuda_stream
        clk = torch.zeros(2, dtype=torch.int64, device="cuda")
        run = lambda: lib.omnitok_debug_mfma_peak(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()),  # noqa
                                                  blocks, iters, lds, ctypes.c_void_p(clk.data_ptr()), s)
        for ms_target in range(2):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            run()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        flops = blocks * 4 * iters * 32 * 4096.0
        c = clk.cpu()
        ghz = float(c[0]) / max(float(c[1]), 1.0) * 0.1
        print(f"{fill:6s} {label:14s} {ms:8.3f} ms  {flops / ms / 1e9:8.2f} TF   shader clock {ghz:.3f} GHz")

