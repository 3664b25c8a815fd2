#!/usr/bin/env python
"""K-step timeline of the persistent GEMM (workgroup 0): s_memtime stamps at the start and the end of
every K-step stream (2 per K-step per wave); deltas alternate stream / gap."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omnitokenizer_amd import _lib, ops
lib = _lib.load()
L, D = 163840, 512
x = torch.randn(L, D, device="cuda")
w = torch.randn(D, D, device="cuda") * 0.04
for _ in range(3):
    ops.linear(x, w)
tr = torch.zeros(8 * 96, dtype=torch.int64, device="cuda")
lib.omnitok_debug_set_gemm_trace(ctypes.c_void_p(tr.data_ptr()))
ops.linear(x, w)
torch.cuda.synchronize()
lib.omnitok_debug_set_gemm_trace(None)
t = tr.cpu().view(8, 96)
t0 = int(t[t > 0].min())
for wv in range(8):
    row = [int(v) - t0 for v in t[wv] if v > 0]
    d = [row[i + 1] - row[i] for i in range(len(row) - 1)]
    print("wave", wv, "start", row[0], "deltas", d[:36])
