"""GPU debug: where does omnitok_layernorm_prevq differ from layernorm + pre_vq?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from omnitokenizer_amd import ops
torch.manual_seed(0)
n, a, c, D = 3, 64, 5, 512
g0 = torch.Generator().manual_seed(171)
x = (torch.randn(n * a * c, D, generator=g0) * 3 + 0.5).cuda()
gam = (torch.randn(D, generator=g0) * 0.3 + 1).cuda()
bet = (torch.randn(D, generator=g0) * 0.2).cuda()
w = (torch.randn(8, D, generator=g0) * 0.05).cuda()
wb = (torch.randn(8, generator=g0) * 0.1).cuda()
def cmp(tag, got, want):
    d = (got != want)
    print(f"{tag}: {int(d.sum())} of {d.numel()} elements differ, max abs {float((got - want).abs().max()):.3e}, rows {int(d.any(1).sum())}", flush=True)
for tr in (False, True):
    for l2 in (False, True):
        for beta in (bet, None):
            ln = ops.layernorm_transposed(x, gam, beta, n, a, c) if tr else ops.layernorm(x, gam, beta)
            cmp(f"tr={tr} l2={l2} beta={beta is not None}", ops.layernorm_prevq(x, gam, beta, w, wb, n, a, c, tr, l2), ops.pre_vq(ln, w, wb, l2))
# one-hot probes: z[c] = LN(x)[pos_c] exactly -> isolates the LayerNorm half
for base in (0, 8, 100, 256, 300, 504):
    wo = torch.zeros(8, D, device="cuda")
    for cc in range(8):
        wo[cc, base + cc] = 1.0
    z0 = torch.zeros(8, device="cuda")
    ln = ops.layernorm(x, gam, bet)
    cmp(f"one-hot LN probe at {base}", ops.layernorm_prevq(x, gam, bet, wo, z0, n, a, c, False, False), ln[:, base:base + 8].contiguous())
    cmp(f"   standalone pre_vq same probe", ops.pre_vq(ln, wo, z0, False), ln[:, base:base + 8].contiguous())
# pre_vq half alone against fp64
ln = ops.layernorm(x, gam, bet)
ref = (ln.double() @ w.double().T + wb.double())
print("pre_vq vs fp64 max abs", float((ops.pre_vq(ln, w, wb, False).double() - ref).abs().max()))
print("fused  vs fp64 max abs", float((ops.layernorm_prevq(x, gam, bet, w, wb, n, a, c, False, False).double() - ref).abs().max()))

# which half differs? LN output of the fused kernel via one-hot probes over ALL positions vs layernorm_kernel
ln = ops.layernorm(x, gam, bet)
bad_cols = 0
for base in range(0, D, 8):
    wo = torch.zeros(8, D, device="cuda")
    for cc in range(8):
        wo[cc, base + cc] = 1.0
    got = ops.layernorm_prevq(x, gam, bet, wo, torch.zeros(8, device="cuda"), n, a, c, False, False)
    bad_cols += int((got != ln[:, base:base + 8]).sum())
print("LN half, all 512 columns probed: mismatching elements", bad_cols, "of", ln.numel())
xa = torch.ones(n * a * c, D, device="cuda"); xa[:, ::2] = -1.0
lnA = ops.layernorm(xa, gam, bet)
gotA = ops.layernorm_prevq(xa, gam, bet, w, wb, n, a, c, False, False)
cmp("alternating +-1 rows", gotA, ops.pre_vq(lnA, w, wb, False))
