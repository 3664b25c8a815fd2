"""Which rows of the LayerNorm-epilogue GEMM differ between the tail schedule (cfg 0) and one launch (cfg 1 / 6 / 7)?"""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from omnitokenizer_amd import ops, _lib

def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()

n_cu = torch.cuda.get_device_properties(0).multi_processor_count
K = 512
for M2 in (128 * (n_cu + 3) - 11, 128 * (n_cu + 3), 128 * n_cu + 32, 128 * n_cu + 64 * 3):
    x, w = rnd(M2, K, seed=80), rnd(512, K, seed=81, scale=0.05)
    bias, res = rnd(512, seed=82), rnd(M2, 512, seed=83)
    gamma, beta = 1.0 + 0.2 * rnd(512, seed=84), 0.1 * rnd(512, seed=85)
    bound = 1.01 * (math.sqrt(512) * float(gamma.abs().max()) + float(beta.abs().max()))
    ap2, asc2 = ops.pl_pack_rows(x)
    wp = ops.pl_pack_weight(w)
    ref = x.double() @ w.double().t() + bias.double() + res.double()
    kw = dict(a_scale=asc2, bias=bias, residual=res, epilogue=2, out_bound=bound, ln=(gamma, beta, 1e-5))
    outs = {}
    for cfg in (1, 0, 6, 7, 0, 1):
        poison = torch.full((M2 * 512 + 4096,), float("nan"), device="cuda"); del poison
        poison = torch.full((((M2 + 255) // 256 * 256) * 512,), -1, device="cuda", dtype=torch.int32); del poison
        c, lp = ops.linear_pl(ap2, wp, M2, 512, K, cfg=cfg, **kw)
        torch.cuda.synchronize()
        err = (c.double() - ref).abs().max(dim=1).values
        bad = torch.nonzero(~(err < 1e-4)).flatten()
        pl = ops.pl_unpack_planes(lp, M2, 512)
        key = f"{cfg}"
        msg = f"M={M2} cfg={cfg}: bad rows {bad.numel()}" + (f" first {bad[:6].tolist()} last {bad[-3:].tolist()}" if bad.numel() else "")
        if key in outs:
            msg += f" | same as first run of cfg {cfg}: c {torch.equal(c, outs[key][0])} planes {torch.equal(pl, outs[key][1])}"
        else:
            outs[key] = (c.clone(), pl.clone())
        if "1" in outs and cfg != 1:
            dc = torch.nonzero((c != outs["1"][0]).any(1)).flatten()
            dp = torch.nonzero((pl != outs["1"][1]).any(1)).flatten()
            msg += f" | vs cfg 1: c rows differ {dc.numel()} {dc[:4].tolist()} planes rows differ {dp.numel()} {dp[:4].tolist()}"
        print(msg, flush=True)
