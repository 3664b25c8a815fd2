"""GPU box: where do the wrong 16-byte units of gemm_plt<7>'s planes come from?  Every wrong unit is looked up among the CORRECT
units of the neighbouring row blocks; prints the distribution of (source location - destination location)."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from omnitokenizer_amd import _lib, ops  # noqa: E402
from tests.test_gpu_temporal_fused import build_operands  # noqa: E402

heads, nseq = 8, 8192
oi, ri = build_operands(ops, nseq, heads, seed=3 + nseq, heavy=True)
pl, sc, st = ops.stats_pack_temporal(oi["x"], nseq)
vb = 1.01 * float(ri["x"].norm(dim=1).max()) * float(ri["wv"].norm(dim=1).max())
args = (pl, sc, st, nseq, heads, oi["wqk"], oi["wv"], oi["fold_qk"], oi["fu_v"], oi["qs"], oi["ks"], 8.0, vb)
_lib.set_option("temporal_kernel", 2)
P0, O0, S0 = ops.temporal_fused(*args)
_lib.set_option("temporal_kernel", 1)
P, O, S = ops.temporal_fused(*args)
good = O0.cpu().numpy().view(np.uint8).reshape(-1, 16)
bad = O.cpu().numpy().view(np.uint8).reshape(-1, 16)
wrong = np.nonzero((good != bad).any(1))[0]
print("wrong 16-byte units", len(wrong), "of", len(good))


def loc(u):  # unit index -> (row block, k block, plane, chunk, row)
    return u // (16 * 512), u // 512 % 16, u // 256 % 2, u // 64 % 4, u % 64


index = {}
for u in range(len(good)):
    index.setdefault(good[u].tobytes(), []).append(u)
rel = collections.Counter()
zero = 0
for u in wrong[:4000]:
    v = bad[u].tobytes()
    if not any(v):
        zero += 1
        continue
    src = index.get(v)
    if not src:
        rel["not found"] += 1
        continue
    d = loc(u)
    best = min(src, key=lambda s: abs(s - u))
    s = loc(best)
    rel[tuple(int(a) - int(b) for a, b in zip(s, d))] += 1
print("all-zero units", zero)
for k, v in rel.most_common(30):
    print("  source - destination (row block, k block, plane, chunk, row):", k, v)
cs = {k: collections.Counter() for k in ("wm", "wn", "ni", "c", "hi", "plane", "r32", "step", "row", "j", "tile%8")}
for u in wrong:
    rb, kb, plane, chunk, row = (int(x) for x in loc(u))
    m = rb * 64 + row
    tile, rr = m // 320, m % 320
    wm, r = rr // 160, rr % 160
    for k, v in (("wm", wm), ("wn", kb % 4 // 2), ("ni", kb % 2), ("c", chunk // 2), ("hi", chunk % 2), ("plane", plane), ("r32", r // 5), ("step", r % 5),
                 ("row", row), ("j", r // 64), ("tile%8", tile % 8)):
        cs[k][v] += 1
for k, v in cs.items():
    print(k, sorted(v.items()))
g16 = torch.from_numpy(good.copy()).view(torch.float16).view(-1, 8).float()
b16 = torch.from_numpy(bad.copy()).view(torch.float16).view(-1, 8).float()
w = torch.from_numpy(wrong)
hi_units = w[(w // 256 % 2) == 0]
print("hi-plane units: max |good|", float(g16[hi_units].abs().max()), "max |bad - good|", float((b16[hi_units] - g16[hi_units]).abs().max()),
      "median |bad - good|", float((b16[hi_units] - g16[hi_units]).abs().max(1).values.median()))
for u in hi_units[:6].tolist():
    print(loc(u), "good", g16[u].tolist(), "bad", b16[u].tolist())
