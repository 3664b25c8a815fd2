"""GPU (-m gpu): the split-operand GEMMs that carry the engine's nn.Linear layers -- gemm_x3 (fp32 operands as
3 bf16 planes, six bf16-MFMA products) and gemm_h2 (2 fp16 planes, three fp16-MFMA products, range-scaled) --
against fp64, against the fp32-MFMA kernel's error, and for the properties the engine relies on: results
independent of the tile shape / problem size (bitwise), determinism across persistent tile switches, rigorous
range bounds from the row-statistics pass, and the three engine modes against the golden fixtures."""
import ctypes
import math

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import GoldenCase

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "needs the MI355X"
    from omnitokenizer_amd import ops as _ops
    return _ops


def dev(t):
    return t.contiguous().cuda()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def maxerr(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item()


@pytest.fixture(params=[0, 1, 3, 4], ids=["auto", "256x256", "128x128", "64-wide"])
def tile(request):
    from omnitokenizer_amd import _lib
    _lib.set_option("x3_tile", request.param)
    _lib.set_option("h2_tile", request.param)
    yield request.param
    _lib.set_option("x3_tile", 0)
    _lib.set_option("h2_tile", 0)


SHAPES = [(1024, 512, 512), (1000, 192, 512), (4096, 1536, 512), (257, 768, 1408), (20000 + 77, 1024, 512),
          (300, 64, 32)]


def _ref(a, w, bias, res):
    ref = a.double() @ w.double().t()
    if bias is not None:
        ref = ref + bias.double()
    if res is not None:
        ref = ref + res.double()
    return ref


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("mode", ["plain", "bias_residual"])
@pytest.mark.parametrize("kind", ["x3", "h2"])
def test_split_gemm_error_class(ops, tile, kind, M, N, K, mode):
    """Same tolerance against fp64 as the fp32-MFMA kernel (tests/test_gpu_ops.py::test_gemm), and never more
    than 2x that kernel's own error: the split GEMMs are fp32-class, not reduced precision."""
    a, w = rnd(M, K, seed=4), rnd(N, K, seed=5) * 0.05
    a[:, :3] *= 40.0          # outlier channels
    bias = rnd(N, seed=6) if mode != "plain" else None
    res = rnd(M, N, seed=7) if mode != "plain" else None
    ref = _ref(a, w, bias, res)
    kw = dict(bias=None if bias is None else dev(bias), residual=None if res is None else dev(res))
    if kind == "x3":
        out = ops.linear_x3(dev(a), dev(w), **kw)
    else:
        out = ops.linear_h2(dev(a), ops.h2_pack_weight(dev(w)), float(a.abs().max()), **kw)
    scale = ref.abs().max().item()
    err = maxerr(out, ref)
    assert err < 3e-6 * max(scale, 1.0) * math.sqrt(K / 512), f"err {err} scale {scale}"
    if mode == "plain":
        err32 = maxerr(ops.linear(dev(a), dev(w)), ref)
        assert err < 2 * err32 + 1e-7, f"{kind} err {err} vs fp32-MFMA err {err32}"


@pytest.mark.parametrize("amp", [1e-6, 1e-3, 1.0, 1e3, 1e6])
@pytest.mark.parametrize("slack", [1.0, 37.0, 4096.0])
def test_h2_range_scaling(ops, amp, slack):
    """fp16 has 5 exponent bits: whatever the magnitude of A and however loose the (valid) bound, the result
    keeps fp32-class relative accuracy."""
    M, N, K = 1500, 512, 512
    a, w = rnd(M, K, seed=1) * amp, rnd(N, K, seed=2) * 0.05
    w[:5] *= 100.0
    w[5:9] *= 1e-4
    ref = a.double() @ w.double().t()
    out = ops.linear_h2(dev(a), ops.h2_pack_weight(dev(w)), float(a.abs().max()) * slack)
    err32 = maxerr(ops.linear(dev(a), dev(w)), ref)
    assert torch.isfinite(out).all()
    assert maxerr(out, ref) < 2 * err32 + 1e-9 * amp


def test_h2_pack_weight(ops):
    w = rnd(300, 512, seed=3) * torch.logspace(-4, 3, 300).view(-1, 1)
    w[7] = 0.0
    planes, scale = ops.h2_pack_weight(dev(w))
    # [N/64 blocks, K/32, plane, k-group, row, 8] -> [rows, K] per plane
    pl = planes.cpu().double().permute(2, 0, 4, 1, 3, 5).reshape(2, -1, 512)
    assert pl[:, 300:].abs().max() == 0, "padding rows of the last block must be zero"
    pl = pl[:, :300]
    rec = (pl[0] + pl[1]) * scale.cpu().double().view(-1, 1)
    assert ((rec - w.double()).abs() <= 2.0 ** -21 * w.double().abs() + 1e-30).all()
    m, e = torch.frexp(scale.cpu())
    assert (m == 0.5).all(), "row scales must be powers of two"
    top = pl[0].abs().max(1).values
    ok = (top >= 2.0 ** 13) & (top <= 2.0 ** 14)
    ok[7] = True
    assert ok.all()


@pytest.mark.parametrize("kind", ["x3", "h2"])
def test_split_gemm_geglu(ops, tile, kind):
    inner, D, pad, M = 1365, 512, 1408, 1000
    x = rnd(M, D, seed=11)
    w1 = rnd(2 * inner, D, seed=12) * 0.05
    val, gate = (x.double() @ w1.double().t()).chunk(2, dim=-1)
    hid_ref = F.gelu(gate) * val
    wp = ops.pack_geglu_weight(dev(w1), pad)
    if kind == "x3":
        hid = ops.linear_x3(dev(x), wp, geglu=True)
    else:
        hid = ops.linear_h2(dev(x), ops.h2_pack_weight(wp), float(x.abs().max()), geglu=True)
    assert hid.shape == (M, pad)
    assert maxerr(hid[:, :inner], hid_ref) < 1e-5 * max(1.0, hid_ref.abs().max().item())
    assert hid[:, inner:].abs().max().item() == 0.0


@pytest.mark.parametrize("kind", ["x3", "h2"])
@pytest.mark.parametrize("ln_cols", [512, 1536])
def test_split_gemm_fused_layernorm(ops, tile, kind, ln_cols):
    """LayerNorm applied while the A tile is staged: columns [0, ln_cols) from LN(x), the rest from x itself
    (Q from LN(x), K/V from x: reference attention.py:404-412)."""
    M, K = 2500, 512
    x = rnd(M, K, seed=21) * 2 + 0.3
    gam, bet = rnd(K, seed=22) * 0.2 + 1, rnd(K, seed=23) * 0.1
    w = rnd(1536, K, seed=24) * 0.05
    xd, gd, bd, wd = dev(x), dev(gam), dev(bet), dev(w)
    st = ops.row_stats(xd)
    y = ops.layernorm(xd, gd, bd)          # the standalone LN kernel (itself tested against the oracle)
    ref = torch.cat([y.double().cpu() @ w[:ln_cols].double().t(), x.double() @ w[ln_cols:].double().t()], 1)
    if kind == "x3":
        out = ops.linear_x3(xd, wd, ln=(st, gd, bd), ln_cols=ln_cols)
    else:
        lnb = math.sqrt(K) * float(gam.abs().max()) + float(bet.abs().max())
        out = ops.linear_h2(xd, ops.h2_pack_weight(wd), float(x.abs().max()), ln=(st, gd, bd), ln_cols=ln_cols,
                            ln_bound=lnb)
    assert maxerr(out, ref) < 2e-5


@pytest.mark.parametrize("kind", ["x3", "h2"])
def test_split_gemm_is_tile_and_size_independent(ops, kind):
    """Bitwise: the rows of a big problem equal the same rows computed alone with another tiling (what makes the
    engine's results independent of the batch size)."""
    from omnitokenizer_amd import _lib
    x, w = dev(rnd(8192, 512, seed=31)), dev(rnd(512, 512, seed=32) * 0.05)
    pk = ops.h2_pack_weight(w)
    f = (lambda a: ops.linear_x3(a, w)) if kind == "x3" else (lambda a: ops.linear_h2(a, pk, 6.0))
    outs = []
    for t in (1, 3, 4):
        _lib.set_option("x3_tile", t)
        _lib.set_option("h2_tile", t)
        outs.append(f(x))
        small = f(x[:100].contiguous())
        assert torch.equal(small, outs[-1][:100])
    _lib.set_option("x3_tile", 0)
    _lib.set_option("h2_tile", 0)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("kind", ["x3", "h2"])
@pytest.mark.parametrize("t,N,geglu", [(3, 2816, True), (3, 1536, False), (1, 2816, True)])
def test_split_gemm_deterministic_across_tile_switches(ops, kind, t, N, geglu):
    """Regression: persistent workgroups walk several tiles; the LayerNorm statistics must travel with each
    staged K-step across the tile switch (they once were cached per tile and went stale for 16-lane groups)."""
    from omnitokenizer_amd import _lib
    M, K = 20480, 512
    x = dev(rnd(M, K, seed=41))
    w = dev(rnd(N, K, seed=42) * 0.05)
    gam, bet = dev(rnd(K, seed=43) * 0.2 + 1), dev(rnd(K, seed=44) * 0.1)
    st = ops.row_stats(x)
    lnb = math.sqrt(K) * float(gam.abs().max()) + float(bet.abs().max())
    pk = ops.h2_pack_weight(w)
    _lib.set_option("x3_tile", t)
    _lib.set_option("h2_tile", t)
    try:
        if kind == "x3":
            f = lambda: ops.linear_x3(x, w, geglu=geglu, ln=(st, gam, bet))  # noqa: E731
        else:
            f = lambda: ops.linear_h2(x, pk, 8.0, geglu=geglu, ln=(st, gam, bet), ln_bound=lnb)  # noqa: E731
        first = f()
        y = ops.layernorm(x, gam, bet)
        rows = torch.arange(0, M, 37, device="cuda")
        h = y[rows].double() @ w.double().t()
        if geglu:
            hb = h.view(len(rows), -1, 2, 32)
            h = (F.gelu(hb[:, :, 1]) * hb[:, :, 0]).reshape(len(rows), -1)
        assert maxerr(first[rows], h) < 3e-5
        for _ in range(15):
            assert torch.equal(f(), first)
    finally:
        _lib.set_option("x3_tile", 0)
        _lib.set_option("h2_tile", 0)


def test_split_gemm_row_map(ops):
    """A rows gathered as groups (frame-0 / rest-frames selection of to_pixels) through the C ABI."""
    from omnitokenizer_amd import _lib
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    K, N = 512, 192
    big, w = rnd(4 * 160, K, seed=9), rnd(N, K, seed=2) * 0.05
    bd, wd = dev(big), dev(w)
    rows = torch.tensor([(m // 64) * 160 + 32 + m % 64 for m in range(256)])
    ref = big[rows].double() @ w.double().t()
    out = torch.empty(256, N, device="cuda")
    assert lib.omnitok_gemm_x3(p(bd), K, p(wd), K, None, None, 0, p(out), N, 256, N, K, 0, 64, 160, 32, None, None,
                               None, 0, None, 0, 0, s) == 0
    assert maxerr(out, ref) < 1e-5
    planes, scale = ops.h2_pack_weight(wd)
    out2 = torch.empty(256, N, device="cuda")
    assert lib.omnitok_gemm_h2(p(bd), K, p(planes), p(scale), None, None, 0, p(out2), N, 256, N, K, 0, 64, 160, 32,
                               float(big.abs().max()), None, 1, 0, None, None, None, 0, 0.0, None, 0, 0, s) == 0
    assert maxerr(out2, ref) < 1e-5


@pytest.mark.parametrize("kind", ["x3", "h2"])
def test_split_gemm_two_output_tensors(ops, tile, kind):
    """Columns [split_col, N) of one launch land in a second dense tensor (Q and K|V of the merged projection)."""
    from omnitokenizer_amd import _lib
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    M, K, N = 1000, 512, 1536
    x, w = dev(rnd(M, K, seed=71)), dev(rnd(N, K, seed=72) * 0.05)
    q = torch.full((M, 512), float("nan"), device="cuda")
    kv = torch.full((M, 1024), float("nan"), device="cuda")
    if kind == "x3":
        whole = ops.linear_x3(x, w)
        assert lib.omnitok_gemm_x3(p(x), K, p(w), K, None, None, 0, p(q), 512, M, N, K, 0, 0, 0, 0, None, None, None, 0,
                                   p(kv), 1024, 512, s) == 0
    else:
        planes, scale = ops.h2_pack_weight(w)
        whole = ops.linear_h2(x, (planes, scale), 6.0)
        assert lib.omnitok_gemm_h2(p(x), K, p(planes), p(scale), None, None, 0, p(q), 512, M, N, K, 0, 0, 0, 0, 6.0, None,
                                   1, 0, None, None, None, 0, 0.0, p(kv), 1024, 512, s) == 0
    assert torch.equal(q, whole[:, :512]) and torch.equal(kv, whole[:, 512:])


@pytest.mark.parametrize("rows_per_clip", [0, 64, 256, 1000])
def test_row_stats_and_ranges(ops, rows_per_clip):
    """(mean, rstd) like LayerNorm; the per-clip ranges are rigorous upper bounds (never below the true max)
    and not absurdly loose."""
    M, D = 2000, 512
    x = rnd(M, D, seed=51) * torch.logspace(-2, 2, M).view(-1, 1) + rnd(M, 1, seed=52) * 5
    x[17] = 0.0
    xd = dev(x)
    n_clips = 1 if rows_per_clip <= 0 else -(-M // rows_per_clip)
    bounds = torch.zeros(n_clips, 2, device="cuda")
    st = ops.row_stats(xd, bounds=bounds, rows_per_clip=rows_per_clip).cpu()
    mean = x.double().mean(1)
    var = x.double().var(1, unbiased=False)
    assert (st[:, 0].double() - mean).abs().max() < 1e-5 * max(1.0, mean.abs().max().item())
    assert ((st[:, 1].double() - 1 / (var + 1e-5).sqrt()).abs() / (1 / (var + 1e-5).sqrt())).max() < 1e-5
    b = bounds.cpu().double()
    rpc = M if rows_per_clip <= 0 else rows_per_clip
    for c in range(n_clips):
        xc = x[c * rpc:(c + 1) * rpc].double()
        amax, nmax = xc.abs().max().item(), xc.norm(dim=1).max().item()
        assert b[c, 0] >= amax and b[c, 1] >= nmax * (1 - 1e-6)
        assert b[c, 0] <= 40 * amax + 0.1 and b[c, 1] <= 1.01 * nmax + 0.1


def test_h2_per_clip_ranges_make_rows_batch_independent(ops):
    """A clip's rows get the same bits whether or not a clip with 1000x larger activations shares the launch."""
    rpc, K, N = 256, 512, 512
    a = rnd(4 * rpc, K, seed=61)
    a[rpc:2 * rpc] *= 1000.0
    w = dev(rnd(N, K, seed=62) * 0.05)
    pk = ops.h2_pack_weight(w)

    def run(x):
        xd = dev(x)
        b = torch.zeros(x.shape[0] // rpc, 2, device="cuda")
        ops.row_stats(xd, bounds=b, rows_per_clip=rpc)
        return ops.linear_h2(xd, pk, 1.01, a_bound_dev=b, a_bound_stride=2, rows_per_clip=rpc)

    full = run(a)
    alone = run(a[:rpc])
    assert torch.equal(full[:rpc], alone)
    assert maxerr(full, a.double() @ w.double().cpu().t()) < 3e-6 * 1000 * 6


@pytest.mark.parametrize("mode", [0, 1, 2], ids=["fp32_mfma", "bf16x3", "fp16x2"])
@pytest.mark.parametrize("name", ["s2_sdpa_r64_vid", "s1_legacy_r64_img", "s2_sdpa_r128_vid_16k", "s2_sdpa_r256_img"])
def test_engine_gemm_modes_vs_golden(mode, name):
    """encode ids bit-exact and decode pixels within 1e-4 of the reference in every GEMM mode of the engine."""
    from omnitokenizer_amd import OmniTokenizer_VQGAN, _lib
    c = GoldenCase(name)
    _lib.set_option("gemm_mode", mode)
    try:
        m = OmniTokenizer_VQGAN(c.args, attention_mode=c.mode)
        m.load_state_dict(c.sd, strict=True)
        m = m.cuda().eval()
        ids, z = m.encode(c.x.cuda(), c.is_image, return_latents=True)
        rec = m.decode(c.ids.cuda(), c.is_image)
        assert int((ids.cpu() != c.ids).sum()) == 0
        assert (z.cpu() - c.z).abs().max().item() < 2e-5
        assert (c.strided(rec.cpu()) - c.recon).abs().max().item() < 1e-4
    finally:
        _lib.set_option("gemm_mode", 2)
