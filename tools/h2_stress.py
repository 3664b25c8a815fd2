#!/usr/bin/env python
"""Determinism / race hunt for the h2 GEMM: repeats each launch and compares bitwise with the first result and
against fp64."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from omnitokenizer_amd import _lib, ops  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)  # noqa: E731
bad = 0
for tile in (0, 1, 3, 4):
    _lib.set_option("h2_tile", tile)
    for (M, N, K, geglu, ln) in ((20480, 512, 512, False, False), (20480, 1536, 512, False, True), (20480, 2816, 512, True, True),
                                 (20480, 512, 1408, False, False), (5120, 512, 512, False, False), (10240, 1536, 512, False, True),
                                 (40960, 512, 512, False, False), (1024, 512, 512, False, False), (20480, 768, 512, False, False)):
        x = r(M, K)
        w = r(N, K) * 0.05
        res = r(M, N // 2 if geglu else N)
        pk = ops.h2_pack_weight(w)
        gam, bet = r(K) * 0.2 + 1, r(K) * 0.1
        st = ops.row_stats(x) if K <= 1024 else None
        lnb = math.sqrt(K) * float(gam.abs().max()) + float(bet.abs().max())
        kw = dict(geglu=geglu)
        if ln and K <= 512:
            kw.update(ln=(st, gam, bet), ln_cols=(512 if N == 1536 else N), ln_bound=lnb)
        if not geglu:
            kw.update(residual=res)
        first = ops.linear_h2(x, pk, 8.0, **kw)
        torch.cuda.synchronize()
        nd = 0
        for it in range(20):
            o = ops.linear_h2(x, pk, 8.0, **kw)
            if not torch.equal(o, first):
                nd += 1
                d = (o - first).abs()
                idx = d.flatten().argmax().item()
                if nd == 1:
                    rows = (d.max(1).values > 0).nonzero().flatten()
                    cols = (d.max(0).values > 0).nonzero().flatten()
                    print(f"   diff max {d.max().item():.3e} at row {idx // o.shape[1]} col {idx % o.shape[1]}; rows {rows[:6].tolist()}..{rows[-1].item()} "
                          f"({len(rows)}), cols {cols[:6].tolist()}..{cols[-1].item()} ({len(cols)})")
        bad += nd
        print(f"tile {tile} M{M} N{N} K{K} geglu {geglu} ln {ln}: nondeterministic repeats {nd}/20", flush=True)
_lib.set_option("h2_tile", 0)
print("TOTAL nondeterministic:", bad)
