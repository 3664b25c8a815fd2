"""GPU box: the fused temporal stage on gemm_plt_kernel ("temporal_kernel" 1) against the same stage on gemm_pl_kernel (0):
same operands, P / planes / scales compared bit for bit, and the launch times of both."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from omnitokenizer_amd import _lib, ops  # noqa: E402
from tests.test_gpu_temporal_fused import build_operands  # noqa: E402

for nseq, heavy in ((1024, True), (4096, False), (16384, False)):
    heads = 8
    oi, ri = build_operands(ops, nseq, heads, seed=3 + nseq, heavy=heavy)
    pl, sc, st = ops.stats_pack_temporal(oi["x"], nseq)
    vb = 1.01 * float(ri["x"].norm(dim=1).max()) * float(ri["wv"].norm(dim=1).max())
    out = {}
    for kern in (0, 1, 2, 12):
        _lib.set_option("temporal_kernel", kern % 10)
        _lib.set_option("pl_stagger", 1 if kern >= 10 else 0)
        args = (pl, sc, st, nseq, heads, oi["wqk"], oi["wv"], oi["fold_qk"], oi["fu_v"], oi["qs"], oi["ks"], 8.0, vb)
        P, planes, osc = ops.temporal_fused(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ops.temporal_fused(*args)
        torch.cuda.synchronize()
        out[kern] = (P.clone(), planes.clone(), osc.clone(), (time.perf_counter() - t0) / 10 * 1e6)
    _lib.set_option("temporal_kernel", 1)
    _lib.set_option("pl_stagger", 0)
    print(f"   one workgroup per CU with skewed waves: max P diff to gemm_pl {float((out[12][0] - out[0][0]).abs().max()):.2e}")
    bad = ((out[1][0] - out[0][0]).abs() > 1e-4).view(nseq, heads, 5, 8).cpu()
    if bad.any():
        import collections
        idx = bad.nonzero()
        print("   wrong P entries", len(idx), "of", bad.numel())
        for name, key in (("tile", idx[:, 0] // 64), ("half", idx[:, 0] % 64 // 32), ("lane", idx[:, 0] % 32), ("head", idx[:, 1]), ("step", idx[:, 2]), ("slot", idx[:, 3])):
            c = collections.Counter(key.tolist())
            print("    by", name, sorted(c.items())[:40])
        # are the scores of a wrong (sequence, head) wrong for every step?  print two of them
        for r in idx[:2]:
            sq, hd = int(r[0]), int(r[1])
            print("    seq", sq, "head", hd, "gemm_pl", out[0][0].view(nseq, heads, 40)[sq, hd].cpu().tolist())
            print("    seq", sq, "head", hd, "gemm_plt", out[1][0].view(nseq, heads, 40)[sq, hd].cpu().tolist())
    a, b = out[0], out[1]
    eqP, eqO, eqS = torch.equal(a[0], b[0]), torch.equal(a[1], b[1]), torch.equal(a[2], b[2])
    dP = float((a[0] - b[0]).abs().max())
    nz = int((a[1] != b[1]).sum()) if not eqO else 0
    print(f"nseq {nseq} heavy {heavy}: P equal {eqP} (max diff {dP:.2e}), planes equal {eqO} ({nz} bytes differ), scales equal {eqS}; "
          f"both launches {a[3]:.0f} us (gemm_pl) vs {b[3]:.0f} us (gemm_plt); one workgroup per CU: P equal to gemm_plt's {torch.equal(out[2][0], b[0])}, "
          f"max diff to gemm_pl {float((out[2][0] - a[0]).abs().max()):.2e}, {out[2][3]:.0f} us", flush=True)
