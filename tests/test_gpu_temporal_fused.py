"""GPU (-m gpu): the fused temporal stage (round 5) -- omnitok_stats_pack_temporal + omnitok_pl_gemm epilogues 6 (q|k GEMM ->
softmax weights) and 7 (V GEMM -> attention output planes) -- against an fp64 restatement of the reference's temporal attention
(attention.py:402-486 with is_spatial = False: Q from LN(x), K / V from the raw x, l2norm, learned scales, SDPA scale 8,
is_causal) and against the unfused kernels (plane GEMM N = 1536 + attn_temporal) it replaces."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float64) * scale


@pytest.fixture(scope="module")
def ops():
    from omnitokenizer_amd import ops as o
    return o


def build_operands(ops, nseq, heads, seed, heavy=False):
    D = heads * 64
    x = rnd(nseq * 5, D, seed=seed) * (3.0 if heavy else 1.0) + 0.3
    if heavy:
        x[:, 7] *= 20.0
    gamma, beta = 1.0 + 0.2 * rnd(D, seed=seed + 1), 0.1 * rnd(D, seed=seed + 2)
    wq, wk, wv = (rnd(D, D, seed=seed + 3 + i, scale=0.05) for i in range(3))
    qs, ks = 1.0 + 0.3 * rnd(64, seed=seed + 6), 1.0 + 0.3 * rnd(64, seed=seed + 7)
    f32 = lambda t: t.float().cuda()  # noqa: E731
    # the engine's folding (engine_build.hip fold_ln_weight + regroup_qk_kernel), restated here
    wq_f = wq * gamma[None, :]
    fb_q = wq @ beta
    fu_k, fu_v = wk.sum(1), wv.sum(1)
    rows, fb, fu = [], [], []
    for h in range(heads):
        rows += [wq_f[h * 64:(h + 1) * 64], wk[h * 64:(h + 1) * 64]]
        fb += [fb_q[h * 64:(h + 1) * 64], torch.zeros(64, dtype=torch.float64)]
        fu += [torch.zeros(64, dtype=torch.float64), fu_k[h * 64:(h + 1) * 64]]
    wqk = torch.cat(rows)
    ops_in = dict(x=f32(x), wqk=ops.pl_pack_weight(f32(wqk)), wv=ops.pl_pack_weight(f32(wv)), fold_qk=(f32(torch.cat(fb)), f32(torch.cat(fu))),
                  fu_v=f32(fu_v), qs=f32(qs), ks=f32(ks))
    ref_in = dict(x=x.float().double(), gamma=gamma, beta=beta, wq=wq.float().double(), wk=wk.float().double(), wv=wv.float().double(),
                  qs=qs.float().double(), ks=ks.float().double())
    # the folded fp32 weights are what the kernel sees: restate the reference on them (gamma folded in fp32, then exact)
    ref_in["wq_folded"] = wq_f.float().double()
    ref_in["fb"] = fb_q.float().double()
    ref_in["fu_k"], ref_in["fu_v"] = fu_k.float().double(), fu_v.float().double()
    return ops_in, ref_in


def test_stats_pack_temporal_is_the_permuted_stats_pack(ops):
    nseq, D = 200, 512  # ragged: 4 tiles of 64 sequences, the last one with 8 live sequences
    x = rnd(nseq * 5, D, seed=1).float().cuda() * torch.logspace(-1, 1, nseq * 5).cuda()[:, None]
    pl, sc, st = ops.stats_pack(x, center=True)
    tpl, tsc, tst = ops.stats_pack_temporal(x, nseq)
    rows = (nseq + 63) // 64 * 320
    p = torch.arange(rows)
    tile, rem = p // 320, p % 320
    seq = tile * 64 + (rem // 160) * 32 + rem % 32
    t = (rem % 160) // 32
    live = seq < nseq
    src = (seq * 5 + t)[live].cuda()
    want = ops.pl_unpack_planes(pl, nseq * 5, D)
    got = ops.pl_unpack_planes(tpl, rows, D)
    assert torch.equal(got[live.cuda()], want[src])
    assert (got[~live.cuda()] == 0).all()
    assert torch.equal(tsc[live.cuda()], sc[src]) and torch.equal(tst[live.cuda()], st[src])
    assert torch.isfinite(tsc).all() and torch.isfinite(tst).all()


@pytest.mark.parametrize("nseq,heavy,alibi", [(64, False, False), (1024, False, False), (1024, True, False), (200, False, True),
                                              (4096, False, False), (8192, True, False)])
def test_fused_temporal_stage_vs_fp64_and_unfused(ops, nseq, heavy, alibi):
    heads, D = 8, 512
    oi, ri = build_operands(ops, nseq, heads, seed=10 + nseq, heavy=heavy)
    slopes = torch.tensor([2.0 ** -(i + 1) for i in range(heads)], dtype=torch.float64) if alibi else None
    pl, sc, st = ops.stats_pack_temporal(oi["x"], nseq)
    v_bound = 1.01 * float(ri["x"].norm(dim=1).max()) * float(ri["wv"].norm(dim=1).max())
    P, oplanes, oscale = ops.temporal_fused(pl, sc, st, nseq, heads, oi["wqk"], oi["wv"], oi["fold_qk"], oi["fu_v"], oi["qs"],
                                            oi["ks"], 8.0, v_bound, alibi=slopes.float().cuda() if alibi else None)
    got = ops.pl_unpack_planes(oplanes, nseq * 5, D) * oscale.double()[:, None]
    # fp64 reference on the fp32 weights the kernels see (LayerNorm folded: rstd ((x - mean) . (Wq o gamma)^T) + Wq beta)
    x = ri["x"]
    mean = x.mean(1, keepdim=True)
    rstd = 1.0 / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-5)
    xc = x - mean
    q = (rstd * (xc @ ri["wq_folded"].T) + ri["fb"]).view(nseq, 5, heads, 64).transpose(1, 2)
    k = (xc @ ri["wk"].T + mean * ri["fu_k"]).view(nseq, 5, heads, 64).transpose(1, 2)
    v = (xc @ ri["wv"].T + mean * ri["fu_v"]).view(nseq, 5, heads, 64).transpose(1, 2)
    q = F.normalize(q, dim=-1) * ri["qs"]
    k = F.normalize(k, dim=-1) * ri["ks"]
    s = 8.0 * (q @ k.transpose(-1, -2))
    if alibi:
        i = torch.arange(5)
        s = s - slopes.view(1, heads, 1, 1) * (i[:, None] - i[None, :]).abs().double()
    s = s.masked_fill(torch.triu(torch.ones(5, 5, dtype=torch.bool), 1), float("-inf"))
    p = s.softmax(-1)
    want = (p @ v).transpose(1, 2).reshape(nseq * 5, D)
    vmax = float(v.abs().max())
    err = float((got.cpu() - want).abs().max())
    # the weights: e / sum e against the fp64 softmax
    Pc = P.cpu().double().view(nseq, heads, 5, 8)
    pk = Pc[..., :5] * Pc[..., 5:6]
    perr = float((pk - p).abs().max())
    print(f"nseq {nseq} heavy {heavy} alibi {alibi}: |out - fp64| {err:.2e} (|v| max {vmax:.1f}), |P - fp64| {perr:.2e}")
    assert perr < 2e-5 and err < 4e-6 * max(1.0, vmax)
    assert (Pc[..., 6:] == 0).all()
    # against the unfused kernels on the same operand: plane GEMM (N = 1536) + attn_temporal
    from omnitokenizer_amd import ops as o2
    xpl, xsc, xst = o2.stats_pack(oi["x"], center=True)
    wall = torch.cat([ri["wq_folded"], ri["wk"], ri["wv"]]).float().cuda()
    fb_all = torch.cat([ri["fb"], torch.zeros(2 * D, dtype=torch.float64)]).float().cuda()
    fu_all = torch.cat([torch.zeros(D, dtype=torch.float64), ri["fu_k"], ri["fu_v"]]).float().cuda()
    qkv = o2.linear_pl(xpl, o2.pl_pack_weight(wall), nseq * 5, 3 * D, D, a_scale=xsc, fold=(xst, fb_all, fu_all, D))
    out = o2.attn_temporal(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], nseq, 5, heads, oi["qs"], oi["ks"], True,
                           alibi=slopes.float().cuda() if alibi else None)
    derr = float((got - out.double()).abs().max())
    print(f"   vs the unfused kernels: {derr:.2e}")
    assert derr < 4e-6 * max(1.0, vmax)


@pytest.mark.parametrize("nseq", [4096, 8192, 16384, 8200])
def test_two_workgroups_per_cu_are_bit_identical_to_one(ops, nseq):
    """gemm_plt_kernel with two workgroups per CU ("temporal_kernel" 1, the default) against the same kernel with one per CU (2),
    ten launches each size, bit for bit: 1, 2 and 4 tiles per workgroup and a ragged last tile.  Builds whose P . V arithmetic the
    compiler had packed into v_pk_*_f32 failed exactly this, every launch, and only with two workgroups per CU
    (profiles/r05_temporal_plt.txt); the first form ("temporal_kernel" 0) bounds both to rounding."""
    from omnitokenizer_amd import _lib
    heads = 8
    oi, ri = build_operands(ops, nseq, heads, seed=3 + nseq, heavy=True)
    pl, sc, st = ops.stats_pack_temporal(oi["x"], nseq)
    vb = 1.01 * float(ri["x"].norm(dim=1).max()) * float(ri["wv"].norm(dim=1).max())
    args = (pl, sc, st, nseq, heads, oi["wqk"], oi["wv"], oi["fold_qk"], oi["fu_v"], oi["qs"], oi["ks"], 8.0, vb)
    try:
        _lib.set_option("temporal_kernel", 2)
        P0, O0, S0 = ops.temporal_fused(*args)
        _lib.set_option("temporal_kernel", 0)
        P1, O1, S1 = ops.temporal_fused(*args)
        _lib.set_option("temporal_kernel", 1)
        for rep in range(10):
            P, O, S = ops.temporal_fused(*args)
            assert torch.equal(P, P0) and torch.equal(O, O0) and torch.equal(S, S0), f"launch {rep}"
    finally:
        _lib.set_option("temporal_kernel", 1)
    assert float((P0 - P1).abs().max()) < 2e-6 and torch.equal(S0, S1)
    a = ops.pl_unpack_planes(O0, nseq * 5, heads * 64) * S0[:, None]
    b = ops.pl_unpack_planes(O1, nseq * 5, heads * 64) * S1[:, None]
    assert float((a - b).abs().max()) < 4e-6 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("name", ["s2_sdpa_r256_vid17", "heavy_s2_sdpa_r256_vid17", "heavy_s2_sdpa_r256_vid17_b8"])
def test_engine_with_the_fused_temporal_stage_vs_reference_golden(name):
    """Whole encode / decode with "temporal_fused" 1 (the default since r05) and 0 (q|k|v GEMM + attn_temporal kernel) against the
    reference's golden outputs: same bars for both, and the two forms agree to rounding."""
    from omnitokenizer_amd import OmniTokenizer_VQGAN, _lib
    from tests.helpers import GoldenCase
    from tests.test_gpu_e2e import PIXEL_TOL, Z_TOL, assert_ids_match_or_near_tie
    c = GoldenCase(name)
    m = OmniTokenizer_VQGAN(c.args, attention_mode=c.mode)
    m.load_state_dict(c.sd, strict=True)
    m = m.cuda().eval()
    try:
        _lib.set_option("temporal_fused", 0)
        ids0, z0 = m.encode(c.x.cuda(), False, return_latents=True)
        _lib.set_option("temporal_fused", 1)
        ids1, z1 = m.encode(c.x.cuda(), False, return_latents=True)
        rec1 = m.decode(c.ids.cuda(), False)
        _lib.set_option("temporal_kernel", 2)   # one workgroup per CU: bit-identical to the default (two)
        ids2, z2 = m.encode(c.x.cuda(), False, return_latents=True)
        _lib.set_option("temporal_kernel", 0)   # the first fused form: same bars
        ids3, z3 = m.encode(c.x.cuda(), False, return_latents=True)
    finally:
        _lib.set_option("temporal_fused", 1)
        _lib.set_option("temporal_kernel", 1)
    assert torch.equal(z1, z2) and torch.equal(ids1, ids2)
    # the option is live (the two forms round differently) -- wherever the call runs the plane flow the fused stage belongs to: always
    # by default; under OMNITOK_TEST_PL_MIN_TOKENS=12288 (the A/B arm of tests/conftest.py) a one-clip call takes the other flow
    import os
    if c.ids.numel() >= int(os.environ.get("OMNITOK_TEST_PL_MIN_TOKENS", "0")):
        assert not torch.equal(z0, z1)
    noise = max(c.fp32_noise_z, 0.0)
    ztol = max(Z_TOL, 8.0 * noise)
    zerr, zdiff = float((z1.cpu() - c.z).abs().max()), float((z1 - z0).abs().max())
    assert zerr < ztol and zdiff < 2 * ztol, (zerr, zdiff, ztol)
    assert float((z3.cpu() - c.z).abs().max()) < ztol and float((z3 - z1).abs().max()) < 2 * ztol
    assert_ids_match_or_near_tie(ids1, c.ids, z1, c.sd["codebook.embeddings"], name + " (fused temporal stage)")
    perr = float((c.strided(rec1.cpu()) - c.recon).abs().max())
    assert perr < max(PIXEL_TOL, 8.0 * c.fp32_noise_pix)
