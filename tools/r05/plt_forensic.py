"""GPU box: what IS the wrong value?  For the wrong plane units of a failing gemm_plt<7> build (two workgroups per CU), solve the
attention output of time step 1 for the v_1 element the kernel must have used and compare it with candidate explanations."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from omnitokenizer_amd import _lib, ops  # noqa: E402
from tests.test_gpu_temporal_fused import build_operands  # noqa: E402

heads, nseq, D = 8, 8192, 512
heavy = len(sys.argv) < 2 or sys.argv[1] != "light"
oi, ri = build_operands(ops, nseq, heads, seed=3 + nseq, heavy=heavy)
pl, sc, st = ops.stats_pack_temporal(oi["x"], nseq)
vb = 1.01 * float(ri["x"].norm(dim=1).max()) * float(ri["wv"].norm(dim=1).max())
args = (pl, sc, st, nseq, heads, oi["wqk"], oi["wv"], oi["fold_qk"], oi["fu_v"], oi["qs"], oi["ks"], 8.0, vb)
_lib.set_option("temporal_kernel", 2)
P0, O0, S0 = ops.temporal_fused(*args)
_lib.set_option("temporal_kernel", 1)
P, O, S = ops.temporal_fused(*args)
print("heavy", heavy, "P equal", torch.equal(P, P0), "scales equal", torch.equal(S, S0))
og = (ops.pl_unpack_planes(O0, nseq * 5, D).double() * S0.double()[:, None]).cpu().view(nseq, 5, D)
ob = (ops.pl_unpack_planes(O, nseq * 5, D).double() * S.double()[:, None]).cpu().view(nseq, 5, D)
bad = (og != ob)
idx = bad.nonzero()
print("wrong output values", len(idx), "steps", sorted(set(idx[:, 1].tolist())), "channels % 64", sorted(set((idx[:, 2] % 64).tolist())))
# fp64 v (same folding as the kernel) and its K-step partial sums
x = ri["x"]
mean = x.mean(1, keepdim=True)
xc = x - mean
Pc = P.cpu().double().view(nseq, heads, 5, 8)
seen = 0
for sq, stp, ch in idx.tolist():
    if stp != 1:
        continue
    hd = ch // 64
    e10, e11, il = Pc[sq, hd, 1, 0], Pc[sq, hd, 1, 1], Pc[sq, hd, 1, 5]
    w = ri["wv"][ch]
    rows = xc[sq * 5: sq * 5 + 5]
    v = rows @ w + mean[sq * 5: sq * 5 + 5, 0] * ri["fu_v"][ch]
    v1_bad = (ob[sq, 1, ch] / il - e10 * v[0]) / e11
    parts = (rows[1].view(32, 16) * w.view(32, 16)).sum(1)   # K-step contributions to xc_1 . w
    cand = {"v1 good": v[1], "xc1.w (no mean term)": rows[1] @ w, "v0": v[0], "v2": v[2], "v3": v[3], "v4": v[4],
            "v1 elem 3": rows[1] @ ri["wv"][ch + 1] + mean[sq * 5 + 1, 0] * ri["fu_v"][ch + 1]}
    miss = v[1] - parts
    j = int((miss - v1_bad).abs().argmin())
    dbl = v[1] + parts
    j2 = int((dbl - v1_bad).abs().argmin())
    print(f"seq {sq} ch {ch}: v1 used {float(v1_bad):+.5f}; " + "; ".join(f"{k} {float(val):+.5f}" for k, val in cand.items()) +
          f"; best 'missing K step' {j}: {float(miss[j]):+.5f}; best 'doubled K step' {j2}: {float(dbl[j2]):+.5f}")
    # with v1 = 0: which v2, v3, v4 did steps 2..4 use?
    used = [float(v[0]), float(v1_bad)]
    for i in (2, 3, 4):
        acc_ = sum(Pc[sq, hd, i, j] * used[j] for j in range(i))
        used.append(float((ob[sq, i, ch] / Pc[sq, hd, i, 5] - acc_) / Pc[sq, hd, i, i]))
    print("      used v0..v4", " ".join(f"{u:+.5f}" for u in used), "| good", " ".join(f"{float(u):+.5f}" for u in v))
    for i in (1, 2, 3, 4):
        terms = [float(Pc[sq, hd, i, j] * v[j]) for j in range(i + 1)]
        want = sum(terms)
        got = float(ob[sq, i, ch] / Pc[sq, hd, i, 5])
        print(f"      step {i}: sum_j e_ij v_j good {want:+.5f} got {got:+.5f} residual {got - want:+.5f} | terms", " ".join(f"{t:+.5f}" for t in terms),
              f"| good output {float(og[sq, i, ch]):+.5f} bad {float(ob[sq, i, ch]):+.5f}")
    seen += 1
    if seen >= 6:
        break
