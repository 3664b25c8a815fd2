"""GPU box: repeat the fused temporal stage on gemm_plt_kernel and compare every repetition with the first form's outputs."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from omnitokenizer_amd import _lib, ops  # noqa: E402
from tests.test_gpu_temporal_fused import build_operands  # noqa: E402

heads = 8
for nseq in (8192, 4096, 16384):
    oi, ri = build_operands(ops, nseq, heads, seed=3 + nseq, heavy=True)
    pl, sc, st = ops.stats_pack_temporal(oi["x"], nseq)
    vb = 1.01 * float(ri["x"].norm(dim=1).max()) * float(ri["wv"].norm(dim=1).max())
    args = (pl, sc, st, nseq, heads, oi["wqk"], oi["wv"], oi["fold_qk"], oi["fu_v"], oi["qs"], oi["ks"], 8.0, vb)
    _lib.set_option("temporal_kernel", 2)
    P0, O0, S0 = ops.temporal_fused(*args)
    _lib.set_option("temporal_kernel", 1)
    nbadP = nbadO = 0
    for rep in range(30):
        P, O, S = ops.temporal_fused(*args)
        bp = (P != P0).view(nseq, heads, 5, 8)
        bo = O != O0
        if bp.any() and nbadP < 3:
            idx = bp.nonzero().cpu()
            print(f"nseq {nseq} rep {rep}: {len(idx)} wrong P entries")
            for name, key in (("tile", idx[:, 0] // 64), ("half", idx[:, 0] % 64 // 32), ("lane", idx[:, 0] % 32), ("head", idx[:, 1]),
                              ("step", idx[:, 2]), ("slot", idx[:, 3])):
                print("    by", name, sorted(collections.Counter(key.tolist()).items())[:24])
        if bo.any() and nbadO < 3:
            idx = bo.view(-1).nonzero().view(-1).cpu()
            # plane byte offset -> (row block of 64, k block of 32, plane, 1 KiB chunk, row, byte)
            rb, kb = idx // (16 * 8192), idx // 8192 % 16
            print(f"nseq {nseq} rep {rep}: {len(idx)} wrong plane bytes; row blocks {sorted(collections.Counter(rb.tolist()).items())[:12]} "
                  f"k blocks {sorted(collections.Counter(kb.tolist()).items())} plane {sorted(collections.Counter((idx // 4096 % 2).tolist()).items())} "
                  f"chunk {sorted(collections.Counter((idx // 1024 % 4).tolist()).items())} rows {sorted(collections.Counter((idx // 16 % 64).tolist()).items())[:70]}")
        nbadP += bool(bp.any())
        nbadO += bool(bo.any())
    print(f"nseq {nseq}: repetitions with wrong P {nbadP} / 30, with wrong planes {nbadO} / 30", flush=True)
