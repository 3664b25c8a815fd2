"""PL_ROWLN at the C3 call size (163840 x 512 x 512): launch time with / without the residual operand, by tile configuration, next to
the plain fp32 epilogue of the same shape -- how much of the launch is the epilogue's residual round trips."""
import math
import sys
import torch
sys.path.insert(0, ".")
from omnitokenizer_amd import ops, _lib  # noqa: E402

M, K = 163840, 512
g = torch.Generator().manual_seed(1)
x = torch.randn(M, K, generator=g).cuda()
w = (0.05 * torch.randn(512, K, generator=g)).cuda()
bias = torch.randn(512, generator=g).cuda()
res = (2.0 * torch.randn(M, 512, generator=g)).cuda()
gamma = (1.0 + 0.2 * torch.randn(512, generator=g)).cuda()
beta = (0.1 * torch.randn(512, generator=g)).cuda()
ap, asc = ops.pl_pack_rows(x)
wp = ops.pl_pack_weight(w)
bound = 1.01 * (math.sqrt(512) * float(gamma.abs().max()) + float(beta.abs().max()))


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


cfgs = [int(c) for c in sys.argv[1:]] or [1, 6]
for cfg in cfgs:
    t_r = timeit(lambda: ops.linear_pl(ap, wp, M, 512, K, a_scale=asc, bias=bias, residual=res, epilogue=2, out_bound=bound,
                                       ln=(gamma, beta, 1e-5), cfg=cfg))
    t_n = timeit(lambda: ops.linear_pl(ap, wp, M, 512, K, a_scale=asc, bias=bias, epilogue=2, out_bound=bound,
                                       ln=(gamma, beta, 1e-5), cfg=cfg))
    print(f"ROWLN cfg {cfg}: with residual {t_r:7.1f} us   without {t_n:7.1f} us")
t_f = timeit(lambda: ops.linear_pl(ap, wp, M, 512, K, a_scale=asc, bias=bias, residual=res, cfg=1))
t_fn = timeit(lambda: ops.linear_pl(ap, wp, M, 512, K, a_scale=asc, bias=bias, cfg=1))
print(f"F32 epilogue cfg 1 (256 x 256 tiles): with residual {t_f:7.1f} us   without {t_fn:7.1f} us")
