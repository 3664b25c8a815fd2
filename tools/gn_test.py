import sys, os
sys.path.insert(0, os.getcwd())
import torch
from omnitokenizer_amd import _lib, ops
from tools.x3_bench import timeit
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
L, D = 163840, 512
x = r(L, D); h = r(L, 1408)
pff = ops.h2_pack_weight(ops.pack_geglu_weight(r(2730, D) * 0.04, 1408))
pkv = ops.h2_pack_weight(r(2 * D, D) * 0.04)
pfo = ops.h2_pack_weight(r(D, 1408) * 0.04)
for gn in (8,):
    _lib.set_option("gemm_gn", gn)
    a = timeit(lambda: ops.linear_h2(x, pff, 8.0, geglu=True), 10)
    b = timeit(lambda: ops.linear_h2(x, pkv, 8.0), 10)
    c = timeit(lambda: ops.linear_h2(h, pfo, 8.0), 10)
    print(f"gn {gn:2d}: ff_in {a:.3f} ms {2.0*L*D*2730/a/1e9:.0f} TF | kv {b:.3f} ms {2.0*L*D*1024/b/1e9:.0f} TF | ff_out {c:.3f} ms {2.0*L*D*1365/c/1e9:.0f} TF")
_lib.set_option("gemm_gn", 8)
st = None
b = torch.zeros(32, 8, 2, device="cuda")
ms = timeit(lambda: ops.row_stats(x, bounds=b, rows_per_clip=5120), 20)
print(f"row_stats with per-clip ranges: {ms:.3f} ms {L*D*4/ms/1e6:.0f} GB/s")
ms = timeit(lambda: ops.row_stats(x), 20)
print(f"row_stats without ranges: {ms:.3f} ms {L*D*4/ms/1e6:.0f} GB/s")
b1 = torch.zeros(1, 8, 2, device="cuda")
ms = timeit(lambda: ops.row_stats(x, bounds=b1, rows_per_clip=0), 20)
print(f"row_stats one clip: {ms:.3f} ms")
y = torch.empty_like(x)
ms = timeit(lambda: ops.layernorm(x, x[0].contiguous()), 20)
print(f"layernorm: {ms:.3f} ms")
