#!/usr/bin/env python
"""Code-range split sweep of the VQ nearest-code search (csrc/vq.hip, option "vq_split") at the BASELINE sizes:
C3 (163 840 rows x 8192 codes), C2 (65 536 x 8192), C5 shape (69 632 x 16 384), one image (1024 x 8192).
Times the whole omnitok_vq_argmin call (fill + sweep + finalize launches, as the engine runs it); 0 = automatic rule."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from omnitokenizer_amd import _lib, ops  # noqa: E402


def timeit(fn, iters=30, warm=3):
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


g = torch.Generator(device="cuda").manual_seed(0)
for name, rows, codes in (("C3", 163840, 8192), ("C2", 65536, 8192), ("C5", 69632, 16384), ("C5x8", 8 * 69632, 16384),
                          ("img", 1024, 8192), ("clip", 5120, 8192)):
    z = torch.nn.functional.normalize(torch.randn(rows, 8, device="cuda", generator=g), dim=1)
    E = torch.randn(codes, 8, device="cuda", generator=g)
    prep = ops.vq_prepare(E)
    ref = None
    line = []
    for rep in range(2):
        for split in (0, 1, 2, 4, 8, 16):
            _lib.set_option("vq_split", split)
            ids = ops.vq_argmin(z, E, prep)
            if ref is None:
                ref = ids
            assert torch.equal(ids, ref)
            ms = timeit(lambda: ops.vq_argmin(z, E, prep))
            line.append(f"split {split}: {ms * 1e3:7.1f} us {2.0 * rows * codes * 8 / ms / 1e9:5.1f} TF")
        print(f"{name:5s} rows {rows} codes {codes} rep {rep}: " + " | ".join(line[-6:]), flush=True)
    _lib.set_option("vq_split", 0)
