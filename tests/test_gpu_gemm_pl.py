"""GPU (-m gpu): the plane x plane GEMM (csrc/gemm_pl.h) that carries to_out / proj, FF-in and FF-out of the
Transformer blocks (reference attention.py:153-168, 386-393, 216-252), its three epilogues, and the attention kernels'
plane outputs that feed it -- against fp64 (the oracle for a GEMM is a matrix product), and for the properties the
engine relies on: results independent of the tile configuration and of the batch (bitwise), ragged sizes, rows beyond
M never stored."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "needs the MI355X"
    from omnitokenizer_amd import ops as _ops
    return _ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def maxerr(a, b):
    return (a.double() - b.double()).abs().max().item()


# (M, N, K): production shapes at reduced M, ragged M, N below / not a multiple of the 256-wide tile
SHAPES = [(1024, 512, 512), (1000, 192, 512), (4096 + 33, 1536, 512), (257, 512, 1408), (5, 64, 32), (777, 768, 512)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("cfg", [1, 2, 5, 6, 0], ids=["256x256", "128x256", "128x128", "128x64", "auto"])
def test_pl_gemm_fp32_epilogue_vs_fp64(ops, M, N, K, cfg):
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    ap, asc = ops.pl_pack_rows(x)
    wp = ops.pl_pack_weight(w)
    ref = x.double() @ w.double().t()
    tol = 2e-6 * math.sqrt(K / 512) * max(1.0, ref.abs().max().item())
    assert maxerr(ops.linear_pl(ap, wp, M, N, K, a_scale=asc, cfg=cfg), ref) < tol
    out = ops.linear_pl(ap, wp, M, N, K, a_scale=asc, bias=bias, residual=res, cfg=cfg)
    assert maxerr(out, ref + bias.double() + res.double()) < tol
    # in place (x += ...), as the engine runs FF-out
    r2 = res.clone()
    g_out = ops.linear_pl(ap, wp, M, N, K, a_scale=asc, residual=r2, cfg=cfg)
    assert torch.equal(g_out, ops.linear_pl(ap, wp, M, N, K, a_scale=asc, residual=res, cfg=cfg))


def test_pl_gemm_planes_roundtrip_and_scales(ops):
    # a plane buffer is the fp32 operand to 22 bits: hi + lo reproduces x * scale within 2^-22 of the row maximum
    x = rnd(300, 512, seed=5) * torch.logspace(-3, 3, 300).cuda()[:, None]
    ap, asc = ops.pl_pack_rows(x)
    back = ops.pl_unpack_planes(ap, 300, 512) * asc.double()[:, None]
    rowmax = x.abs().max(dim=1).values.double()
    assert ((back - x.double()).abs().max(dim=1).values <= rowmax * 2.0 ** -21).all()
    # a static bound gives one scale for all rows
    ap2, none = ops.pl_pack_rows(x, static_bound=float(x.abs().max()) * 1.01)
    assert none is None
    back2 = ops.pl_unpack_planes(ap2, 300, 512) * ops.pl_unscale(float(x.abs().max()) * 1.01)
    assert (back2 - x.double()).abs().max().item() <= float(x.abs().max()) * 2.0 ** -21


def test_pl_gemm_results_do_not_depend_on_tiling_or_batch(ops):
    M, N, K = 2048 + 64, 1024, 512
    x, w = rnd(M, K, seed=6), rnd(N, K, seed=7, scale=0.05)
    ap, asc = ops.pl_pack_rows(x)
    wp = ops.pl_pack_weight(w)
    full1 = ops.linear_pl(ap, wp, M, N, K, a_scale=asc, cfg=1)
    for cfg in (2, 5, 6, 0):   # 128x256, 128x128, 128x64 tiles, and the size rule: same bits
        assert torch.equal(full1, ops.linear_pl(ap, wp, M, N, K, a_scale=asc, cfg=cfg)), cfg
    bias, res = rnd(N, seed=70), rnd(M, N, seed=71)
    fr = ops.linear_pl(ap, wp, M, N, K, a_scale=asc, bias=bias, residual=res, cfg=1)
    for cfg in (5, 6):
        assert torch.equal(fr, ops.linear_pl(ap, wp, M, N, K, a_scale=asc, bias=bias, residual=res, cfg=cfg)), cfg
    # the first 512 rows alone (another grid, other tiles per workgroup): bit-identical rows
    ap3, asc3 = ops.pl_pack_rows(x[:512].contiguous())
    assert torch.equal(ops.linear_pl(ap3, wp, 512, N, K, a_scale=asc3, cfg=1), full1[:512])
    # determinism across launches
    assert torch.equal(ops.linear_pl(ap, wp, M, N, K, a_scale=asc, cfg=1), full1)


def test_pl_gemm_tail_schedule_is_bit_identical(ops):
    """r06 tail schedule (gemm_pl.hip pl_auto_plan): when the 256 x 256 tiles make full rounds over the CUs plus a mostly idle last
    round, the size rule runs the full rounds on them and the remaining rows on thin tiles in a second launch.  Same bits as one
    launch of either configuration, for every epilogue; "pl_tail" 0 switches it off."""
    from omnitokenizer_amd import _lib
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    K = 512
    # N = 512: 2 column tiles -> (n_cu / 2 + 1) row tiles = one full round + 2 tiles
    M = 256 * (n_cu // 2 + 1) - 37
    Mx = 128 * (n_cu + 3)                                   # the largest row count used below
    x, w = rnd(Mx, K, seed=80), rnd(512, K, seed=81, scale=0.05)
    bias, res_all = rnd(512, seed=82), rnd(Mx, 512, seed=83)
    res = res_all[:M].contiguous()
    ap, asc = ops.pl_pack_rows(x[:M].contiguous())
    wp = ops.pl_pack_weight(w)
    auto = ops.linear_pl(ap, wp, M, 512, K, a_scale=asc, bias=bias, residual=res, cfg=0)
    assert torch.equal(auto, ops.linear_pl(ap, wp, M, 512, K, a_scale=asc, bias=bias, residual=res, cfg=1))
    assert torch.equal(auto, ops.linear_pl(ap, wp, M, 512, K, a_scale=asc, bias=bias, residual=res, cfg=6))
    try:
        _lib.set_option("pl_tail", 0)
        assert torch.equal(auto, ops.linear_pl(ap, wp, M, 512, K, a_scale=asc, bias=bias, residual=res, cfg=0))
    finally:
        _lib.set_option("pl_tail", 1)
    # LayerNorm epilogue: 128-row tiles, n_cu + 3 of them
    M2 = 128 * (n_cu + 3) - 11
    assert M2 <= Mx
    gamma, beta = 1.0 + 0.2 * rnd(512, seed=84), 0.1 * rnd(512, seed=85)
    bound = 1.01 * (math.sqrt(512) * float(gamma.abs().max()) + float(beta.abs().max()))
    ap2, asc2 = ops.pl_pack_rows(x[:M2].contiguous())
    kw = dict(a_scale=asc2, bias=bias, residual=res_all[:M2].contiguous(), epilogue=2, out_bound=bound, ln=(gamma, beta, 1e-5))
    c0, l0 = ops.linear_pl(ap2, wp, M2, 512, K, cfg=0, **kw)
    c1, l1 = ops.linear_pl(ap2, wp, M2, 512, K, cfg=1, **kw)
    if not torch.equal(c0, c1):   # say which rows, and which of the two is off the fp64 product
        ref2 = x[:M2].double() @ w.double().t() + bias.double() + res_all[:M2].double()
        rows = torch.nonzero((c0 != c1).any(1)).flatten()
        e0, e1 = (c0.double() - ref2).abs().max(1).values, (c1.double() - ref2).abs().max(1).values
        raise AssertionError(f"tail schedule vs one launch: {rows.numel()} rows differ, first {rows[:8].tolist()} last {rows[-4:].tolist()}; "
                             f"max err cfg 0 {float(e0.max()):.3e} (rows > 1e-4: {int((e0 > 1e-4).sum())}), cfg 1 {float(e1.max()):.3e} "
                             f"(rows > 1e-4: {int((e1 > 1e-4).sum())}); options {[(n, _lib.get_option(n)) for n in ('pl_cfg', 'pl_tail', 'gemm_pl', 'gemm_mode')]}")
    assert torch.equal(ops.pl_unpack_planes(l0, M2, 512), ops.pl_unpack_planes(l1, M2, 512))
    # GEGLU: N = 2816 -> 11 column tiles; 24 row tiles = 264 tiles on 256 CUs
    nbn = 11
    M3 = 256 * (n_cu // nbn + 1) - 5
    assert M3 <= Mx
    w1p = ops.pack_geglu_weight(rnd(2 * 1365, K, seed=86, scale=0.05), 1408)
    ap3, _ = ops.pl_pack_rows(x[:M3].contiguous(), static_bound=8.0)
    kw = dict(a_scale_const=ops.pl_unscale(8.0), epilogue=1, out_bound=64.0)
    h0 = ops.linear_pl(ap3, ops.pl_pack_weight(w1p), M3, 2816, K, cfg=0, **kw)
    h1 = ops.linear_pl(ap3, ops.pl_pack_weight(w1p), M3, 2816, K, cfg=1, **kw)
    assert torch.equal(ops.pl_unpack_planes(h0, M3, 1408), ops.pl_unpack_planes(h1, M3, 1408))
    # packed Q | K and V (N = 1024 / 512), whole sequences of 256 tokens
    heads, D, Ntok = 8, 512, 256
    M4 = Ntok * ((256 * (n_cu // 4 + 1)) // Ntok + 1)   # (its own operand below)
    x4 = rnd(M4, D, seed=87, scale=1.5) + 0.2
    wq = rnd(3 * D, D, seed=88, scale=0.05)
    gam4 = 1.0 + 0.2 * rnd(D, seed=89)
    qs, ks = 1.0 + 0.1 * rnd(64, seed=90), 1.0 + 0.1 * rnd(64, seed=91)
    cos, sin = (t.cuda() for t in ops.rope_table(Ntok))
    planes, scales, stats = ops.stats_pack(x4, center=True)
    w2, b, u = ops.fold_layernorm_weight(wq, gam4, None, rows_fold=D)
    attn = dict(n_tokens=Ntok, heads=heads, q_scale=qs, k_scale=ks, cos=cos, sin=sin, q_mul=8.0, q_bound=64.0, k_bound=8.0, v_bound=64.0)
    wpq, wv = ops.pl_pack_weight(w2), ops.pl_pack_weight(w2[2 * D:].contiguous())
    outs = []
    for cfg in (0, 1):
        qp, kp = ops.linear_pl(planes, wpq, M4, 2 * D, D, a_scale=scales, fold=(stats, b, u, D), epilogue=4, attn=attn, cfg=cfg)
        vp = ops.linear_pl(planes, wv, M4, D, D, a_scale=scales, fold=(stats, None, u[2 * D:].contiguous(), 0), epilogue=3, attn=attn, cfg=cfg)
        outs.append((qp, kp, vp))
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)


def test_pl_gemm_two_operands_and_two_outputs(ops):
    # merged launch: columns < 256 from operand a (e.g. LayerNorm(x)), the rest from a2 (raw x); outputs split at 256
    M, K = 1300, 512
    xa, xb = rnd(M, K, seed=8), rnd(M, K, seed=9, scale=3.0)
    w = rnd(768, K, seed=10, scale=0.05)
    ap, asc = ops.pl_pack_rows(xa, static_bound=float(xa.abs().max()) * 1.01)
    bp, bsc = ops.pl_pack_rows(xb)
    wp = ops.pl_pack_weight(w)
    c, c2 = ops.linear_pl(ap, wp, M, 768, K, a_scale_const=ops.pl_unscale(float(xa.abs().max()) * 1.01), a2=(bp, bsc, 0.0),
                          a_split_n=256, c_split_n=256)
    assert maxerr(c, xa.double() @ w[:256].double().t()) < 1e-5
    assert maxerr(c2, xb.double() @ w[256:].double().t()) < 3e-5


def test_pl_gemm_geglu_epilogue_writes_hidden_planes(ops):
    # FF-in (reference attention.py:153-168): value | gate halves, exact-erf GELU; the hidden leaves as planes
    M, K, inner, inner_pad = 1000, 512, 1365, 1408
    x = rnd(M, K, seed=11)
    w1 = rnd(2 * inner, K, seed=12, scale=0.05)
    w1p = ops.pack_geglu_weight(w1, inner_pad)
    ap, _ = ops.pl_pack_rows(x, static_bound=8.0)
    hid = ops.linear_pl(ap, ops.pl_pack_weight(w1p), M, 2 * inner_pad, K, a_scale_const=ops.pl_unscale(8.0), epilogue=1,
                        out_bound=64.0)
    for cfg in (5, 6):   # the small-call tiles write the same planes bit for bit
        hid_c = ops.linear_pl(ap, ops.pl_pack_weight(w1p), M, 2 * inner_pad, K, a_scale_const=ops.pl_unscale(8.0), epilogue=1,
                              out_bound=64.0, cfg=cfg)
        assert torch.equal(ops.pl_unpack_planes(hid_c, M, inner_pad), ops.pl_unpack_planes(hid, M, inner_pad)), cfg
    h = (x.double() @ w1.double().t())
    ref = F.gelu(h[:, inner:]) * h[:, :inner]
    got = ops.pl_unpack_planes(hid, M, inner_pad) * ops.pl_unscale(64.0)
    assert maxerr(got[:, :inner], ref) < 5e-6 * max(1.0, ref.abs().max().item())
    assert (got[:, inner:] == 0).all()  # pad columns are exactly zero (zero weight rows)
    # ... and feed FF-out
    w2 = rnd(512, inner, seed=13, scale=0.05)
    w2p = torch.zeros(512, inner_pad, device="cuda")
    w2p[:, :inner] = w2
    res = rnd(M, 512, seed=14)
    out = ops.linear_pl(hid, ops.pl_pack_weight(w2p), M, 512, inner_pad, a_scale_const=ops.pl_unscale(64.0), residual=res)
    for cfg in (1, 5, 6):
        assert torch.equal(out, ops.linear_pl(hid, ops.pl_pack_weight(w2p), M, 512, inner_pad, a_scale_const=ops.pl_unscale(64.0),
                                              residual=res, cfg=cfg, k_valid=inner)), cfg
    ref2 = ref @ w2.double().t() + res.double()
    assert maxerr(out, ref2) < 1e-5 * max(1.0, ref2.abs().max().item())


@pytest.mark.parametrize("M", [128, 1000, 4096 + 17])
@pytest.mark.parametrize("with_beta", [True, False])
def test_pl_gemm_layernorm_epilogue(ops, M, with_beta):
    # to_out with the residual add and the FeedForward's LayerNorm in the epilogue (reference attention.py:666-680, 163)
    K = 512
    x, w = rnd(M, K, seed=15), rnd(512, K, seed=16, scale=0.05)
    bias, res = rnd(512, seed=17), rnd(M, 512, seed=18, scale=2.0)
    gamma, beta = 1.0 + 0.2 * rnd(512, seed=19), (0.1 * rnd(512, seed=20) if with_beta else None)
    ap, asc = ops.pl_pack_rows(x)
    bound = 1.01 * (math.sqrt(512) * float(gamma.abs().max()) + (float(beta.abs().max()) if with_beta else 0.0))
    c, lnp = ops.linear_pl(ap, ops.pl_pack_weight(w), M, 512, K, a_scale=asc, bias=bias, residual=res, epilogue=2,
                           out_bound=bound, ln=(gamma, beta, 1e-5))
    ref = x.double() @ w.double().t() + bias.double() + res.double()
    assert maxerr(c, ref) < 5e-6 * ref.abs().max().item()
    ln = F.layer_norm(ref, (512,), gamma.double(), beta.double() if with_beta else None, 1e-5)
    got = ops.pl_unpack_planes(lnp, M, 512) * ops.pl_unscale(bound)
    assert maxerr(got, ln) < 1e-5
    # batch independence of the fused statistics: the first 128 rows alone give the same planes and rows
    ap1, asc1 = ops.pl_pack_rows(x[:128].contiguous())
    c1, lnp1 = ops.linear_pl(ap1, ops.pl_pack_weight(w), 128, 512, K, a_scale=asc1, bias=bias, residual=res[:128].contiguous(),
                             epilogue=2, out_bound=bound, ln=(gamma, beta, 1e-5))
    assert torch.equal(c1, c[:128])
    assert torch.equal(ops.pl_unpack_planes(lnp1, 128, 512), ops.pl_unpack_planes(lnp, 128, 512))
    # 128-, 64- and 32-row tiles (r06: small calls) write the same rows and the same LayerNorm planes
    for cfg in (1, 6, 7):
        cc, lc = ops.linear_pl(ap, ops.pl_pack_weight(w), M, 512, K, a_scale=asc, bias=bias, residual=res, epilogue=2,
                               out_bound=bound, ln=(gamma, beta, 1e-5), cfg=cfg)
        assert torch.equal(cc, c), cfg
        assert torch.equal(ops.pl_unpack_planes(lc, M, 512), ops.pl_unpack_planes(lnp, M, 512)), cfg


def test_stats_pack_matches_row_stats_and_pack_rows(ops):
    x = rnd(1000, 512, seed=40) * torch.logspace(-2, 2, 1000).cuda()[:, None] + 0.3
    planes, scales, stats = ops.stats_pack(x)
    ref_planes, ref_scales = ops.pl_pack_rows(x)
    assert torch.equal(scales, ref_scales)
    # centred form: the planes of x - mean (fp32 subtraction of the mean the kernel reports)
    cpl, csc, cst = ops.stats_pack(x, center=True)
    assert torch.equal(cst, stats)
    xc = x - stats[:, :1]
    rpl, rsc = ops.pl_pack_rows(xc.contiguous())
    assert torch.equal(csc, rsc) and torch.equal(ops.pl_unpack_planes(cpl, 1000, 512), ops.pl_unpack_planes(rpl, 1000, 512))
    assert torch.equal(ops.pl_unpack_planes(planes, 1000, 512), ops.pl_unpack_planes(ref_planes, 1000, 512))
    st = ops.row_stats(x)
    assert maxerr(stats[:, 0], st[:, 0]) < 1e-6 * float(x.abs().max()) and ((stats[:, 1] - st[:, 1]).abs() <= 1e-6 * st[:, 1]).all()
    assert (ops.pl_unpack_planes(planes, 1024, 512)[1000:] == 0).all()  # pad rows of the last block are zero


@pytest.mark.parametrize("with_beta", [False, True])
def test_pl_gemm_folded_layernorm(ops, with_beta):
    # merged q | k | v from the planes of the CENTRED rows: Q = LayerNorm(x) . Wq^T (gain folded into the weight, rstd and
    # b = Wq beta in the epilogue), K | V = x . Wkv^T = (x - mean) . Wkv^T + mean u (reference attention.py:404-412);
    # rows with a large mean: no cancellation anywhere
    M, D = 1300, 512
    x = rnd(M, D, seed=41, scale=2.0) + torch.linspace(-3, 3, M).cuda()[:, None]
    w = rnd(3 * D, D, seed=42, scale=0.05)
    gamma = 1.0 + 0.2 * rnd(D, seed=43)
    beta = 0.1 * rnd(D, seed=44) if with_beta else None
    planes, scales, stats = ops.stats_pack(x, center=True)
    w2, b, u = ops.fold_layernorm_weight(w, gamma, beta, rows_fold=D)
    q, kv = ops.linear_pl(planes, ops.pl_pack_weight(w2), M, 3 * D, D, a_scale=scales, fold=(stats, b, u, D), c_split_n=D)
    ln = F.layer_norm(x.double(), (D,), gamma.double(), beta.double() if with_beta else None, 1e-5)
    assert maxerr(q, ln @ w[:D].double().t()) < 2e-5
    assert maxerr(kv, x.double() @ w[D:].double().t()) < 2e-5


@pytest.mark.parametrize("rope", [True, False])
def test_pl_gemm_packs_attention_operands(ops, rope):
    """The q | k launch with the packing epilogue and the swapped-orientation V launch write what attn_pack writes
    (RoPE + l2norm + scales; reference attention.py:417-437): spatial attention on them equals the fp32 route."""
    Bn, N, heads, D = 3, 256, 8, 512
    M = Bn * N
    x = rnd(M, D, seed=45, scale=1.5) + 0.2
    w = rnd(3 * D, D, seed=46, scale=0.05)
    gamma = 1.0 + 0.2 * rnd(D, seed=47)
    qs, ks = 1.0 + 0.1 * rnd(64, seed=48), 1.0 + 0.1 * rnd(64, seed=49)
    cos, sin = (t.cuda() for t in ops.rope_table(N)) if rope else (None, None)
    # fp32 route: LN -> projections -> attn_pack -> attention
    ln = F.layer_norm(x, (D,), gamma, None, 1e-5)
    q = (ln.double() @ w[:D].double().t()).float()
    k = (x.double() @ w[D:2 * D].double().t()).float()
    v = (x.double() @ w[2 * D:].double().t()).float()
    packed, bounds = ops.attn_pack(q, k, v, N, heads, qs, ks, cos=cos, sin=sin)
    o_ref = ops.attn_spatial_h2(packed, bounds, Bn, N, heads)
    # plane route
    planes, scales, stats = ops.stats_pack(x, center=True)
    w2, b, u = ops.fold_layernorm_weight(w, gamma, None, rows_fold=D)
    wp = ops.pl_pack_weight(w2)
    attn = dict(n_tokens=N, heads=heads, q_scale=qs, k_scale=ks, cos=cos, sin=sin, q_mul=8.0, q_bound=bounds[0],
                k_bound=bounds[1], v_bound=bounds[2])
    qp, kp = ops.linear_pl(planes, wp, M, 2 * D, D, a_scale=scales, fold=(stats, b, u, D), epilogue=4, attn=attn)
    wv = ops.pl_pack_weight(w2[2 * D:].contiguous())
    vp = ops.linear_pl(planes, wv, M, D, D, a_scale=scales, fold=(stats, None, u[2 * D:].contiguous(), 0), epilogue=3, attn=attn)

    for cfg in (1, 5, 6):   # every tile configuration packs the same bytes
        q2, k2 = ops.linear_pl(planes, wp, M, 2 * D, D, a_scale=scales, fold=(stats, b, u, D), epilogue=4, attn=attn, cfg=cfg)
        v2 = ops.linear_pl(planes, wv, M, D, D, a_scale=scales, fold=(stats, None, u[2 * D:].contiguous(), 0), epilogue=3, attn=attn,
                           cfg=cfg)
        assert torch.equal(q2, qp) and torch.equal(k2, kp) and torch.equal(v2, vp), cfg

    def unpack(buf):   # hi + lo of every packed element, position by position
        h = buf.view(torch.float16).view(-1, 2, 4096 // 2)  # [32-token block of a head][plane][2048 halfs]
        return h[:, 0].double() + h[:, 1].double()
    for name, got, ref in (("q", qp, packed[0]), ("k", kp, packed[1]), ("v", vp, packed[2])):
        d = (unpack(got) - unpack(ref)).abs().max().item()
        assert d <= 2.0 ** 15 * 2e-5, (name, d)   # operands are scaled to <= 2^15: 2e-5 relative to the bound
    o = ops.attn_spatial_h2(torch.stack([qp, kp, vp]), bounds, Bn, N, heads)
    assert maxerr(o, o_ref) < 3e-5 * float(v.abs().max())


def test_pl_gemm_argument_errors(ops):
    x, w = rnd(64, 512, seed=21), rnd(500, 512, seed=22)
    with pytest.raises(ValueError):
        ops.pl_pack_weight(w)  # N % 32
    ap, asc = ops.pl_pack_rows(x)
    wp = ops.pl_pack_weight(rnd(256, 512, seed=23))
    with pytest.raises(ValueError):  # the LayerNorm epilogue owns whole rows of 512
        ops.linear_pl(ap, wp, 64, 256, 512, a_scale=asc, epilogue=2, out_bound=30.0, ln=(rnd(256), None, 1e-5))
    with pytest.raises(ValueError):  # GEGLU needs the bound of the hidden
        ops.linear_pl(ap, wp, 64, 256, 512, a_scale=asc, epilogue=1)


# ---- attention kernels writing planes ----------------------------------------------------------------------------------

def test_spatial_attention_plane_output_equals_fp32_output(ops):
    Bn, N, heads = 3, 256, 8
    q, k, v = rnd(Bn * N, 512, seed=24), rnd(Bn * N, 512, seed=25), rnd(Bn * N, 512, seed=26, scale=2.0)
    qs, ks = 1.0 + 0.1 * rnd(64, seed=27), 1.0 + 0.1 * rnd(64, seed=28)
    packed, bounds = ops.attn_pack(q, k, v, N, heads, qs, ks)
    o = ops.attn_spatial_h2(packed, bounds, Bn, N, heads)
    planes, scales = ops.attn_spatial_h2_planes(packed, bounds, Bn, N, heads)
    got = ops.pl_unpack_planes(planes, Bn * N, 512) * scales.double()[:, None]
    assert (scales == ops.pl_unscale(bounds[2])).all()
    assert maxerr(got, o) <= bounds[2] * 2.0 ** -21


def test_window_attention_plane_output_equals_fp32_output(ops):
    Bn, gh, gw, heads = 2, 16, 16, 8
    qkv = rnd(Bn * gh * gw, 1536, seed=29)
    bias = rnd(heads, 64, 64, seed=30, scale=0.5)
    o = ops.attn_window(qkv, bias, Bn, gh, gw, heads)
    bound = 1.01 * float(qkv[:, 1024:].abs().max())
    planes = ops.attn_window_planes(qkv, bias, Bn, gh, gw, heads, bound)
    got = ops.pl_unpack_planes(planes, Bn * gh * gw, 512) * ops.pl_unscale(bound)
    assert maxerr(got, o) <= bound * 2.0 ** -21


@pytest.mark.parametrize("T", [1, 2, 5, 9, 17])
@pytest.mark.parametrize("causal", [True, False])
def test_temporal_attention_plane_output_equals_fp32_output(ops, T, causal):
    cols, heads = 48, 8
    q, k, v = rnd(cols * T, 512, seed=31), rnd(cols * T, 512, seed=32), rnd(cols * T, 512, seed=33, scale=2.0)
    qs, ks = 1.0 + 0.1 * rnd(64, seed=34), 1.0 + 0.1 * rnd(64, seed=35)
    o = ops.attn_temporal(q, k, v, cols, T, heads, qs, ks, causal)
    # per-clip bounds: 3 clips of 16 columns
    bdev = torch.tensor([1.0, 9.0, 2.0, 9.0, 4.0, 9.0], device="cuda")
    vb = 1.01 * float(v.abs().max())
    planes, scales = ops.attn_temporal_planes(q, k, v, cols, T, heads, qs, ks, causal, vb, v_bound_dev=bdev, v_bound_stride=2,
                                              cols_per_clip=16)
    got = ops.pl_unpack_planes(planes, cols * T, 512) * scales.double()[:, None]
    assert maxerr(got, o) <= 4 * vb * 2.0 ** -21
    for clip, f in enumerate((1.0, 2.0, 4.0)):
        assert (scales[clip * 16 * T:(clip + 1) * 16 * T] == ops.pl_unscale(vb * f)).all()


def test_engine_plane_data_flow_matches_fp32_data_flow(ops):
    """The engine with gemm_pl 1 (planes between the kernels) against gemm_pl 0 (r02: fp32 activations, split in the K
    loop): the same arithmetic class, so ids agree except provable near-ties and pixels agree far inside the 1e-4 bar."""
    from omnitokenizer_amd import OmniTokenizer_VQGAN, _lib, make_args, synth
    from omnitokenizer_amd.config import OmniTokConfig
    args = make_args(2, resolution=64, sequence_length=5)
    model = OmniTokenizer_VQGAN(args)
    model.load_state_dict(synth.synth_state_dict(OmniTokConfig.from_args(args), seed=3), strict=True)
    model = model.cuda().eval()
    x = synth.synth_video(2, 5, 64, seed=7).cuda()
    try:
        _lib.set_option("gemm_pl", 0)
        ids0 = model.encode(x, False)
        px0 = model.decode(ids0, False)
        _lib.set_option("gemm_pl", 1)
        ids1 = model.encode(x, False)
        px1 = model.decode(ids0, False)
    finally:
        _lib.set_option("gemm_pl", 1)
    assert (ids0 != ids1).float().mean().item() <= 2e-3
    assert maxerr(px0, px1) < 2e-5


def test_window_attention_from_packed_operands_vs_fp64(ops):
    """The r04 window path end to end at the operator level (reference attention.py:254-293): centred planes in window-major
    order -> plane GEMM with the LayerNorm folded into every column and the packing epilogues (no l2norm, q * head_dim^-0.5)
    -> omnitok_attn_window_h2 (dense relative-position bias, output back in token order), against an fp64 restatement of
    LayerNorm -> qkv -> window_partition -> softmax(q k^T scale + bias) v -> window_reverse; fp32 rows and plane output agree."""
    torch.manual_seed(0)
    Bn, gh, gw, heads, D = 2, 16, 24, 8, 512
    rows = Bn * gh * gw
    x = rnd(rows, D, seed=301, scale=1.5) + 0.3
    gamma, beta = rnd(D, seed=302) * 0.1 + 1.0, rnd(D, seed=303) * 0.1
    w = rnd(3 * D, D, seed=304, scale=0.04)
    bias = rnd(heads, 64, 64, seed=305, scale=0.5)   # [head][key][query]
    # fp64 reference
    xd = x.double()
    ln = (xd - xd.mean(1, keepdim=True)) / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + 1e-5) * gamma.double() + beta.double()
    qkv = ln @ w.double().t()

    def part(t):   # [rows, D] -> [Bn, nwy, nwx, 64, heads, 64]
        t = t.reshape(Bn, gh // 8, 8, gw // 8, 8, heads, 64).permute(0, 1, 3, 2, 4, 5, 6)
        return t.reshape(Bn, gh // 8, gw // 8, 64, heads, 64)
    q, k, v = (part(qkv[:, i * D:(i + 1) * D]) for i in range(3))
    s_ = torch.einsum("bijqhd,bijkhd->bijhqk", q * 0.125, k) + bias.double().permute(0, 2, 1)[None, None, None]
    o = torch.einsum("bijhqk,bijkhd->bijqhd", torch.softmax(s_, -1), v)
    ref = o.reshape(Bn, gh // 8, gw // 8, 8, 8, heads * 64).permute(0, 1, 3, 2, 4, 5).reshape(rows, heads * 64)
    # the HIP path
    planes, scales, stats = ops.stats_pack_windows((x), gh, gw, center=True)
    wf, fb, fu = ops.fold_layernorm_weight((w), (gamma), (beta), rows_fold=3 * D)
    wp = ops.pl_pack_weight(wf)
    l2 = float((ln.norm(dim=1).max()) * 1.01)
    qb = l2 * float(w[:D].norm(dim=1).max()) * 0.125
    kb = l2 * float(w[D:2 * D].norm(dim=1).max())
    vb = l2 * float(w[2 * D:].norm(dim=1).max())
    attn = dict(n_tokens=64, heads=heads, q_scale=None, k_scale=None, q_mul=0.125, q_bound=qb, k_bound=kb, v_bound=vb)
    qp, kp = ops.linear_pl(planes, wp, rows, 2 * D, D, a_scale=scales, epilogue=4, fold=(stats, fb, None, 2 * D), attn=attn)
    wv = ops.pl_pack_weight(wf[2 * D:].contiguous())
    vp = ops.linear_pl(planes, wv, rows, D, D, a_scale=scales, epilogue=3, fold=(stats, fb[2 * D:].contiguous(), None, D), attn=attn)
    out = ops.attn_window_h2(qp, kp, vp, (bias), qb, kb, vb, Bn, gh, gw, heads)
    err = maxerr(out, ref)
    print(f"window attention from packed operands: max err {err:.2e} (|ref| max {ref.abs().max():.2f})")
    assert err < 2e-5
    op = ops.attn_window_h2(qp, kp, vp, (bias), qb, kb, vb, Bn, gh, gw, heads, planes=True)
    got = ops.pl_unpack_planes(op, rows, heads * 64) * ops.pl_unscale(vb)
    assert maxerr(got, out) < 2e-6 * max(1.0, float(out.abs().max()))


def test_layernorm_planes_matches_layernorm(ops):
    x = rnd(700, 512, seed=311, scale=2.0) + 0.5
    g, b = rnd(512, seed=312) * 0.1 + 1.0, rnd(512, seed=313) * 0.2
    ref = ops.layernorm(x, g, b)
    bound = float(ref.abs().max()) * 1.01
    got = ops.pl_unpack_planes(ops.layernorm_planes((x), (g), (b), bound), 700, 512) * ops.pl_unscale(bound)
    assert maxerr(got, ref) <= float(ref.abs().max()) * 2.0 ** -21


def test_pl_gemm_k_valid_skips_only_zero_padding(ops):
    """k_valid stops the K loop at the last step that holds non-padding k: with both operands zero beyond it the result is
    bit-identical to the full loop (FF-out: K = 1408 holds 1365 hidden channels)."""
    M, N, K, kv = 777, 512, 1408, 1365
    a, w = rnd(M, K, seed=321), rnd(N, K, seed=322, scale=0.05)
    a[:, kv:] = 0.0
    w[:, kv:] = 0.0
    planes, scales = ops.pl_pack_rows(a)
    wp = ops.pl_pack_weight(w)
    full = ops.linear_pl(planes, wp, M, N, K, a_scale=scales)
    part = ops.linear_pl(planes, wp, M, N, K, a_scale=scales, k_valid=kv)
    assert torch.equal(full, part)
    assert maxerr(part, a.double() @ w.double().t()) < 3e-5


def test_pl_gemm_unpatchify_store_and_operand_row_map(ops):
    """to_pixels as one launch (reference omnitokenizer.py:1006-1017): the tokens of frames 1.. of a [B, T, S] token tensor are read
    through the operand row map and the fp32 result (+ bias) is scattered into the video by the un-patchify epilogue; equal to
    Linear + omnitok_unpatchify on the same rows, frame 0 of the video untouched."""
    B, T, gh, gw, D, C, p, pt = 2, 3, 8, 32, 512, 3, 8, 4
    S = gh * gw
    tok = rnd(B * T * S, D, seed=331)
    w, bias = rnd(C * pt * p * p, D, seed=332, scale=0.05), rnd(C * pt * p * p, seed=333)
    F_, H, W = 1 + (T - 1) * pt, gh * p, gw * p
    bound = float(tok.abs().max()) * 1.01
    planes, _ = ops.pl_pack_rows(tok, static_bound=bound)   # the row map needs ONE static operand scale (rows are re-indexed)
    wp = ops.pl_pack_weight(w)
    video = torch.full((B, C, F_, H, W), 7.0, device="cuda")
    M = B * (T - 1) * S
    ops.linear_pl(planes, wp, M, C * pt * p * p, D, a_scale_const=ops.pl_unscale(bound), bias=bias, epilogue=5,
                  a_rows=((T - 1) * S, T * S, S), unpatch=dict(video=video, f0=1, t=T - 1, pt=pt, p=p))
    with pytest.raises(ValueError):   # per-row scales cannot follow the row map
        ops.linear_pl(planes, wp, M, C * pt * p * p, D, a_scale=torch.ones(B * T * S, device="cuda"), bias=bias, epilogue=5,
                      a_rows=((T - 1) * S, T * S, S), unpatch=dict(video=video.clone(), f0=1, t=T - 1, pt=pt, p=p))
    rows = tok.reshape(B, T, S, D)[:, 1:].reshape(M, D)
    ref_rows = (rows.double() @ w.double().t() + bias.double()).float()
    ref = torch.full_like(video, 7.0)
    ops.unpatchify(ref_rows, ref, 1, T - 1, pt, p)
    assert torch.equal(video[:, :, 0], ref[:, :, 0])          # frame 0 belongs to the other launch
    assert maxerr(video, ref) < 3e-5
    for cfg in (1, 6):   # 256 x 256 and 256 x 64 tiles (r06) scatter the same pixels
        v2 = torch.full_like(video, 7.0)
        ops.linear_pl(planes, wp, M, C * pt * p * p, D, a_scale_const=ops.pl_unscale(bound), bias=bias, epilogue=5, cfg=cfg,
                      a_rows=((T - 1) * S, T * S, S), unpatch=dict(video=v2, f0=1, t=T - 1, pt=pt, p=p))
        assert torch.equal(v2, video), cfg
