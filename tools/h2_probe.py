import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from omnitokenizer_amd import _lib, ops
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
M, K = 512, 512
x = r(M, K) * 2 + 0.3
w = torch.eye(K, device="cuda")
pk = ops.h2_pack_weight(w)
gam, bet = r(K) * 0.2 + 1, r(K) * 0.1
st = ops.row_stats(x)
y = ops.layernorm(x, gam, bet)
lnb = math.sqrt(K) * float(gam.abs().max()) + float(bet.abs().max())
for tile in (3, 1, 4):
    _lib.set_option("h2_tile", tile)
    o = ops.linear_h2(x, pk, float(x.abs().max()), ln=(st, gam, bet), ln_bound=lnb)
    d = (o - y).abs()
    badrows = (d.max(1).values > 1e-4).nonzero().flatten().tolist()
    print("tile", tile, "bad rows", badrows[:40], "n", len(badrows))
    if badrows:
        rr = badrows[0]
        print(" row", rr, "out[:8]", o[rr, :8].tolist())
        print("   expect LN", y[rr, :8].tolist())
        print("   raw x     ", x[rr, :8].tolist())
        print("   stats row", st[rr].tolist(), "mean/rstd true", x[rr].mean().item(), (1 / (x[rr].var(unbiased=False) + 1e-5).sqrt()).item())
        # which row's LN would give this? solve for (m, rs) from two elements
        for cand in range(M):
            yc = (x[rr] - st[cand, 0]) * st[cand, 1] * gam + bet
            if (yc - o[rr]).abs().max() < 1e-3:
                print("   matches LN with stats of row", cand)
                break
        badcols = (d[rr] > 1e-4).nonzero().flatten().tolist()
        print("   bad cols in that row", badcols[:32], len(badcols))
