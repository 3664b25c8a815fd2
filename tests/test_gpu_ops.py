"""GPU (-m gpu): every HIP operator against the CPU oracle / an fp64 torch restatement of the same
op, through the C ABI (omnitokenizer_amd/ops.py -> libomnitok.so).  Tolerances are written per
test: bit-exact for the VQ ids, fp32-roundoff class for everything else (the kernels compute in
fp32 like the reference; only summation order differs)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import c_oracle
from oracle import omnitok_oracle as orc
from tests.helpers import GOLDEN, GoldenCase

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "needs the MI355X"
    from omnitokenizer_amd import ops as _ops
    return _ops


@pytest.fixture(params=[0, 4, 5, 6], ids=["auto_small_tiles", "tile_per_wg", "persistent_128x128", "persistent_256x128"])
def gemm_variant(request):
    """Runs a test once per GEMM kernel variant (4-6: forced whatever the problem size; 0: the
    size-based choice between the 64x128 small-problem kernel and one 128x128 tile per workgroup)."""
    from omnitokenizer_amd import _lib
    _lib.set_option("gemm_variant", request.param)
    yield request.param
    _lib.set_option("gemm_variant", 1)


def dev(t):
    return t.contiguous().cuda()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def maxerr(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item()


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dim", [192, 384, 512, 768])
@pytest.mark.parametrize("with_beta", [True, False])
def test_layernorm(ops, dim, with_beta):
    x = rnd(1001, dim, seed=1) * 3 + 0.7
    g, b = rnd(dim, seed=2) * 0.1 + 1, rnd(dim, seed=3) * 0.1
    ref = orc.layer_norm(x, g, b if with_beta else torch.zeros(dim))
    out = ops.layernorm(dev(x), dev(g), dev(b) if with_beta else None)
    assert maxerr(out, ref) < 2e-5


@pytest.mark.parametrize("M,N,K", [(1024, 512, 512), (1000, 192, 512), (300, 8, 512), (130, 1024, 192),
                                   (4096, 1536, 512), (257, 768, 1408), (20000 + 77, 1024, 512)])
@pytest.mark.parametrize("mode", ["plain", "bias", "residual", "bias_residual", "bias_leaky"])
def test_gemm(ops, gemm_variant, M, N, K, mode):
    a, w = rnd(M, K, seed=4), rnd(N, K, seed=5) * 0.05
    bias = rnd(N, seed=6) if "bias" in mode else None
    res = rnd(M, N, seed=7) if "residual" in mode else None
    ref = a.double() @ w.double().t()
    if bias is not None:
        ref = ref + bias.double()
    if "leaky" in mode:
        ref = F.leaky_relu(ref, 0.1)
    if res is not None:
        ref = ref + res.double()
    out = ops.linear(dev(a), dev(w), None if bias is None else dev(bias), None if res is None else dev(res),
                     leaky="leaky" in mode)
    scale = ref.abs().max().item()
    assert maxerr(out, ref) < 3e-6 * max(scale, 1.0) * math.sqrt(K / 512), f"scale {scale}"


def test_gemm_inplace_residual_and_row_map(ops, gemm_variant):
    import ctypes
    from omnitokenizer_amd import _lib
    M, N, K = 640, 512, 512
    a, w, x = rnd(M, K, seed=1), rnd(N, K, seed=2) * 0.05, rnd(M, N, seed=3)
    xd = dev(x)
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    ad, wd = dev(a), dev(w)
    assert lib.omnitok_gemm(p(ad), K, p(wd), K, None, p(xd), N, p(xd), N, M, N, K, 2, 0, 0, 0, s) == 0
    assert maxerr(xd, a.double() @ w.double().t() + x.double()) < 1e-5
    # A rows gathered as groups: m -> (m // 64) * 160 + 32 + m % 64  (frame-0 / rest-frames selection)
    big = rnd(4 * 160, K, seed=9)
    out = torch.empty(256, N, device="cuda")
    bd = dev(big)
    assert lib.omnitok_gemm(p(bd), K, p(wd), K, None, None, 0, p(out), N, 256, N, K, 0, 64, 160, 32, s) == 0
    rows = torch.tensor([(m // 64) * 160 + 32 + m % 64 for m in range(256)])
    assert maxerr(out, big[rows].double() @ w.double().t()) < 1e-5


@pytest.mark.parametrize("M", [512, 1000])
def test_gemm_geglu_feedforward(ops, gemm_variant, M):
    inner, D, pad = 1365, 512, 1408
    x = rnd(M, D, seed=11)
    w1, w2 = rnd(2 * inner, D, seed=12) * 0.05, rnd(D, inner, seed=13) * 0.05
    val, gate = (x.double() @ w1.double().t()).chunk(2, dim=-1)
    hid_ref = F.gelu(gate) * val
    w1p = ops.pack_geglu_weight(dev(w1), pad)
    hid = ops.linear_geglu(dev(x), w1p)
    assert hid.shape == (M, pad)
    assert maxerr(hid[:, :inner], hid_ref) < 1e-5 * max(1.0, hid_ref.abs().max().item())
    assert hid[:, inner:].abs().max().item() == 0.0
    w2p = torch.zeros(D, pad)
    w2p[:, :inner] = w2
    out = ops.linear(hid, dev(w2p), residual=dev(x))
    assert maxerr(out, hid_ref @ w2.double().t() + x.double()) < 3e-5


@pytest.mark.parametrize("pt,frames", [(1, 1), (4, 9), (2, 5)])
def test_patchify_ln_and_unpatchify(ops, pt, frames):
    B, C, H, W, p = 2, 3, 64, 64, 8
    Fr = 1 + frames if pt > 1 else 1
    video = rnd(B, C, Fr, H, W, seed=21)
    f0, t = (1, (Fr - 1) // pt) if pt > 1 else (0, 1)
    K = C * pt * p * p
    g, b = rnd(K, seed=22) * 0.1 + 1, rnd(K, seed=23) * 0.1
    ref = orc.layer_norm(orc.patchify(video[:, :, f0:f0 + t * pt], p, pt), g, b).reshape(-1, K)
    out = ops.patchify_ln(dev(video), f0, t, pt, p, dev(g), dev(b))
    assert maxerr(out, ref) < 2e-5
    tok = rnd(B * t * (H // p) * (W // p), K, seed=24)
    vid = torch.zeros(B, C, Fr, H, W, device="cuda")
    ops.unpatchify(dev(tok), vid, f0, t, pt, p)
    ref_v = orc.unpatchify(tok.reshape(B, t, H // p, W // p, K), C, p, pt)
    assert torch.equal(vid[:, :, f0:f0 + t * pt].cpu(), ref_v)
    if f0:
        assert vid[:, :, :f0].abs().max().item() == 0.0


@pytest.mark.parametrize("shape", [(2, 5, 8, 8), (1, 1, 32, 32), (3, 3, 16, 16), (1, 17, 8, 8), (1, 2, 64, 64),
                                   (2, 4, 6, 40)])
@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("variant", [0, 2, 3, 1], ids=["regblock", "wide_slab", "lds_ring", "auto"])
def test_peg3d(ops, shape, causal, variant):
    from omnitokenizer_amd import _lib
    _lib.set_option("peg_variant", variant)
    B, T, H, W = shape
    D = 512
    x = rnd(B * T, H * W, D, seed=31)
    w = (torch.rand(D, 1, 3, 3, 3, generator=torch.Generator().manual_seed(32)) * 2 - 1) / math.sqrt(27)
    b = rnd(D, seed=33) * 0.05
    ref = orc.peg(x, w, b, shape, causal) + x
    out = ops.peg3d(dev(x), ops.pack_peg_weight(dev(w)), dev(b), shape, causal)
    assert maxerr(out, ref) < 1e-5
    # the temporal layout is the same raw-buffer view: '(b h w) t d' input
    xt = x.reshape(B, T, H * W, D).permute(0, 2, 1, 3).reshape(B * H * W, T, D).contiguous()
    ref_t = orc.peg(xt, w, b, shape, causal) + xt
    out_t = ops.peg3d(dev(xt), ops.pack_peg_weight(dev(w)), dev(b), shape, causal)
    assert maxerr(out_t, ref_t) < 1e-5


@pytest.mark.parametrize("shape", [(2, 5, 32, 32), (1, 1, 4, 16), (5, 1, 32, 32), (3, 1, 8, 48), (1, 6, 16, 32), (1, 7, 8, 16), (1, 17, 8, 16),
                                   (3, 2, 64, 16)])
@pytest.mark.parametrize("causal", [True, False])
def test_peg3d_wide_slab_kernel_is_bit_identical(ops, shape, causal):
    """csrc/peg_wide.h (64-channel slab, two ring slots, every plane read from LDS once and accumulated into the three output
    planes it feeds) against the time-ring kernel and the oracle on grids it accepts (W % 16 == 0, H % 4 == 0): one plane, plane
    counts inside and outside the window the default rule uses it for, one-tile and multi-tile grids.  Same order of planes and
    taps per output, so bit for bit."""
    from omnitokenizer_amd import _lib
    B, T, H, W = shape
    D = 512
    x = rnd(B * T, H * W, D, seed=35)
    w = (torch.rand(D, 1, 3, 3, 3, generator=torch.Generator().manual_seed(36)) * 2 - 1) / math.sqrt(27)
    b = rnd(D, seed=37) * 0.05
    wp = ops.pack_peg_weight(dev(w))
    try:
        _lib.set_option("peg_variant", 2)
        out = [ops.peg3d(dev(x), wp, dev(b), shape, causal) for _ in range(3)]
        _lib.set_option("peg_variant", 3)
        want = ops.peg3d(dev(x), wp, dev(b), shape, causal)
        _lib.set_option("peg_variant", 1)
        auto = ops.peg3d(dev(x), wp, dev(b), shape, causal)
    finally:
        _lib.set_option("peg_variant", 1)
    assert maxerr(want, orc.peg(x, w, b, shape, causal) + x) < 1e-5
    for o in out + [auto]:
        assert torch.equal(o, want)


def test_fused_token_transposes(ops):
    """LayerNorm / row gather with the '(n a c) d -> (n c a) d' rearrange fused into the store == the plain kernel
    followed by transpose_tokens, bit for bit."""
    n, a, c, D = 3, 5, 64, 512
    x = dev(rnd(n * a * c, D, seed=141))
    g, b = dev(rnd(D, seed=142) * 0.1 + 1), dev(rnd(D, seed=143) * 0.1)
    want = ops.transpose_tokens(ops.layernorm(x, g, b), n, a, c)
    assert torch.equal(ops.layernorm_transposed(x, g, b, n, a, c), want)
    table = dev(rnd(300, D, seed=144))
    ids = dev(torch.randint(0, 300, (n, a, c), generator=torch.Generator().manual_seed(145)))
    want = ops.transpose_tokens(ops.gather_rows(ids, table).reshape(-1, D), n, a, c)
    assert torch.equal(ops.gather_rows(ids, table, transpose=(a, c)).reshape(-1, D), want)


def test_transpose_tokens(ops):
    B, A, C, D = 2, 5, 64, 512
    x = rnd(B, A, C, D, seed=41)
    out = ops.transpose_tokens(dev(x), B, A, C)
    assert torch.equal(out.cpu().reshape(B, C, A, D), x.permute(0, 2, 1, 3).contiguous())


@pytest.mark.parametrize("N", [64, 256, 1024, 4096])
def test_rope_table(ops, N):
    cos, sin = ops.rope_table(N)
    rc, rs = orc.rope_table(N)
    assert maxerr(cos, rc) < 1e-6 and maxerr(sin, rs) < 1e-6


@pytest.mark.parametrize("rope", [True, False])
def test_qk_prep(ops, rope):
    Bn, N, h, d = 3, 64, 8, 64
    q, kv = rnd(Bn * N, h * d, seed=51), rnd(Bn * N, 2 * h * d, seed=52)
    qs, ks = rnd(d, seed=53) * 0.1 + 1, rnd(d, seed=54) * 0.1 + 1
    qr, kr = q.reshape(Bn, N, h, d), kv[:, : h * d].reshape(Bn, N, h, d)
    cos = sin = None
    if rope:
        cos, sin = orc.rope_table(N, d)
        qr, kr = orc.apply_rope(qr, cos, sin), orc.apply_rope(kr, cos, sin)
    qref = orc.l2norm(qr) * qs * 8.0
    kref = orc.l2norm(kr) * ks
    qd, kvd = dev(q), dev(kv)
    ops.qk_prep_(qd, kvd[:, : h * d], N, h, dev(qs), dev(ks), None if cos is None else dev(cos),
                 None if sin is None else dev(sin))
    assert maxerr(qd.reshape(Bn, N, h, d), qref) < 5e-6
    assert maxerr(kvd[:, : h * d].reshape(Bn, N, h, d), kref) < 2e-6
    assert torch.equal(kvd[:, h * d:].cpu(), kv[:, h * d:])  # V untouched


def _ref_attention(q, k, v, bias=None, causal=False):
    s = torch.einsum("bhid,bhjd->bhij", q.double(), k.double())
    if bias is not None:
        s = s + bias.double()
    if causal:
        n = s.shape[-1]
        s = s.masked_fill(torch.ones(n, n, dtype=torch.bool).triu(1), float("-inf"))
    return torch.einsum("bhij,bhjd->bhid", s.softmax(-1), v.double())


@pytest.mark.parametrize("Bn,N", [(3, 64), (2, 192), (1, 1024), (2, 576)])
def test_attn_spatial(ops, Bn, N):
    h, d = 8, 64
    q = orc.l2norm(rnd(Bn, N, h, d, seed=61)) * 8.0
    k = orc.l2norm(rnd(Bn, N, h, d, seed=62))
    kv = torch.cat([k.reshape(Bn * N, h * d), rnd(Bn * N, h * d, seed=63)], dim=1)
    v = kv[:, h * d:].reshape(Bn, N, h, d)
    ref = _ref_attention(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3))
    ref = ref.permute(0, 2, 1, 3).reshape(Bn * N, h * d)
    kvd = dev(kv)
    out = ops.attn_spatial(dev(q.reshape(Bn * N, h * d)), kvd[:, : h * d], kvd[:, h * d:], Bn, N, h)
    assert maxerr(out, ref) < 1e-5


def test_attn_spatial_forced_rescale(ops):
    """A key whose logit towers over the others late in the sweep forces the online-softmax rescale
    branch with a large factor (cdna guide 5.4 rule 26)."""
    Bn, N, h, d = 1, 256, 8, 64
    q = orc.l2norm(rnd(Bn, N, h, d, seed=64)) * 8.0
    k = orc.l2norm(rnd(Bn, N, h, d, seed=65))
    k[0, 200] = q[0, 17] * 4.0  # logit 8*8*4 = 256 for query 17 at key 200
    v = rnd(Bn, N, h, d, seed=66)
    kv = torch.cat([k.reshape(N, h * d), v.reshape(N, h * d)], dim=1)
    ref = _ref_attention(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3))
    ref = ref.permute(0, 2, 1, 3).reshape(N, h * d)
    kvd = dev(kv)
    out = ops.attn_spatial(dev(q.reshape(N, h * d)), kvd[:, : h * d], kvd[:, h * d:], Bn, N, h)
    assert torch.isfinite(out).all()
    assert maxerr(out, ref) < 2e-5


def test_attn_spatial_legacy_bias(ops):
    c = GoldenCase("s1_legacy_r64_img")
    p = "encoder.enc_spatial_transformer.layers.0.1.spatial_rel_pos_bias"
    gh = gw = 8
    Bn, N, h, d = 2, 64, 8, 64
    full = orc.continuous_position_bias(c.sd, p, gh, gw)                  # h, N, N
    tab = orc.continuous_position_bias_table(c.sd, p, gh, gw)             # h, 2gh-1, 2gw-1
    tab_dev = dev(tab.permute(1, 2, 0).reshape(-1, h))                    # [(2gh-1)(2gw-1), h]
    q = orc.l2norm(rnd(Bn, N, h, d, seed=67)) * 8.0
    k = orc.l2norm(rnd(Bn, N, h, d, seed=68))
    v = rnd(Bn, N, h, d, seed=69)
    ref = _ref_attention(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3), bias=full[None])
    ref = ref.permute(0, 2, 1, 3).reshape(Bn * N, h * d)
    kvd = dev(torch.cat([k.reshape(Bn * N, h * d), v.reshape(Bn * N, h * d)], dim=1))
    out = ops.attn_spatial(dev(q.reshape(Bn * N, h * d)), kvd[:, : h * d], kvd[:, h * d:], Bn, N, h, tab_dev, gh, gw)
    assert maxerr(out, ref) < 2e-5


@pytest.mark.parametrize("Bn,g", [(2, 8), (3, 16), (1, 32)])
def test_attn_window(ops, Bn, g):
    c = GoldenCase("s2_sdpa_r64_img")
    p = "encoder.enc_spatial_transformer.layers.2.1"
    N, heads, C = g * g, 8, 512
    qkv = rnd(Bn * N, 3 * C, seed=71)
    table, index = c.sd[f"{p}.relative_position_bias_table"], c.sd[f"{p}.relative_position_index"]
    # oracle restatement of reference attention.py:266-286 on a given qkv
    ws = 8
    def part(t):  # [Bn*N, C] -> windows [BW, 64, heads, 64] -> [BW, heads, 64, 64]
        t = t.reshape(Bn, g // ws, ws, g // ws, ws, heads, 64).permute(0, 1, 3, 2, 4, 5, 6)
        return t.reshape(-1, ws * ws, heads, 64).permute(0, 2, 1, 3)
    q, k, v = (part(qkv[:, i * C:(i + 1) * C]) for i in range(3))
    rpb = table[index.view(-1)].view(64, 64, -1).permute(2, 0, 1)
    ref = _ref_attention(q * 0.125, k, v, bias=rpb[None])
    ref = ref.permute(0, 2, 1, 3).reshape(Bn, g // ws, g // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    ref = ref.reshape(Bn * N, C)
    bias_dense = dev(rpb.permute(0, 2, 1))  # [h, kv, q]
    out = ops.attn_window(dev(qkv), bias_dense, Bn, g, g, heads)
    assert maxerr(out, ref) < 1e-5


@pytest.mark.parametrize("T", [1, 2, 3, 5, 9, 11, 13, 17, 21])
@pytest.mark.parametrize("causal,alibi", [(True, False), (False, False), (True, True)])
def test_attn_temporal(ops, T, causal, alibi):
    cols, h, d = 70, 8, 64
    q, kv = rnd(cols * T, h * d, seed=81), rnd(cols * T, 2 * h * d, seed=82)
    qs, ks = rnd(d, seed=83) * 0.1 + 1, rnd(d, seed=84) * 0.1 + 1
    qq = orc.l2norm(q.reshape(cols, T, h, d)) * qs * 8.0
    kk = orc.l2norm(kv[:, : h * d].reshape(cols, T, h, d)) * ks
    vv = kv[:, h * d:].reshape(cols, T, h, d)
    bias = None
    slopes = None
    if alibi:
        slopes = torch.tensor(orc.alibi_slopes(h), dtype=torch.float32)
        ar = torch.arange(T)
        bias = (-(ar[None, :] - ar[:, None]).abs().float())[None] * slopes.view(h, 1, 1)
        bias = bias[None]
    ref = _ref_attention(qq.permute(0, 2, 1, 3), kk.permute(0, 2, 1, 3), vv.permute(0, 2, 1, 3), bias, causal)
    ref = ref.permute(0, 2, 1, 3).reshape(cols * T, h * d)
    kvd = dev(kv)
    out = ops.attn_temporal(dev(q), kvd[:, : h * d], kvd[:, h * d:], cols, T, h, dev(qs), dev(ks), causal,
                            None if slopes is None else dev(slopes))
    assert maxerr(out, ref) < 1e-5


@pytest.mark.parametrize("shape", [(3, 64, 5), (2, 16, 17), (5, 7, 1), (1, 1, 3)])
@pytest.mark.parametrize("l2", [True, False])
def test_layernorm_prevq_matches_two_pass(ops, shape, l2):
    """The encoder's last LayerNorm fused with pre_vq (omnitok_layernorm_prevq) == layernorm[_transposed] followed by
    pre_vq, bit for bit (with and without the token transpose, ragged row counts, beta = None)."""
    n, a, c = shape
    D = 512
    x = dev(rnd(n * a * c, D, seed=171) * 3 + 0.5)
    g, b = dev(rnd(D, seed=172) * 0.3 + 1), dev(rnd(D, seed=173) * 0.2)
    w, wb = dev(rnd(8, D, seed=174) * 0.05), dev(rnd(8, seed=175) * 0.1)
    for beta in (b, None):
        want = ops.pre_vq(ops.layernorm_transposed(x, g, beta, n, a, c), w, wb, l2)
        got = ops.layernorm_prevq(x, g, beta, w, wb, n, a, c, True, l2)
        assert torch.equal(got, want)
        want = ops.pre_vq(ops.layernorm(x, g, beta), w, wb, l2)
        got = ops.layernorm_prevq(x, g, beta, w, wb, n, a, c, False, l2)
        assert torch.equal(got, want)


def test_pre_vq_and_dequant(ops):
    c = GoldenCase("s2_sdpa_r64_img")
    x = rnd(1000, 512, seed=91)
    w, b = c.sd["pre_vq_conv.1.weight"], c.sd["pre_vq_conv.1.bias"]
    ref = orc.pre_vq(c.sd, x, c.cfg)
    out = ops.pre_vq(dev(x), dev(w), dev(b), True)
    assert maxerr(out, ref) < 2e-6
    raw = ops.pre_vq(dev(x), dev(w), dev(b), False)
    assert maxerr(raw, F.linear(x, w, b)) < 1e-5
    ids = torch.randint(0, 8192, (3, 2, 8, 8), generator=torch.Generator().manual_seed(92))
    E, pw, pb = c.sd["codebook.embeddings"], c.sd["post_vq_conv.1.weight"], c.sd["post_vq_conv.1.bias"]
    tok = ops.dequant_post_vq(dev(ids), dev(E), dev(pw), dev(pb))
    assert maxerr(tok, F.linear(F.embedding(ids, E), pw, pb)) < 1e-5
    with pytest.raises(IndexError):
        bad = ids.clone()
        bad[0, 0, 0, 0] = 8192
        ops.dequant_post_vq(dev(bad), dev(E), dev(pw), dev(pb))
    # the table form the engine decodes with: rows bit-identical to the kernel above, gather exact
    table = ops.dequant_table(dev(E), dev(pw), dev(pb))
    assert tuple(table.shape) == (8192, 512)
    assert torch.equal(ops.gather_rows(dev(ids), table), tok)
    with pytest.raises(IndexError):
        ops.gather_rows(dev(bad), table)
    with pytest.raises(IndexError):
        ops.gather_rows(dev(-bad), table)


def test_torch_custom_ops_dispatch(ops):
    """torch.ops.omnitok.* run the same kernels as the wrappers."""
    g = np.load(os.path.join(GOLDEN, "vq_kat_8192.npz"))
    z, E = dev(torch.from_numpy(g["z"][:512])), dev(torch.from_numpy(g["codebook"]))
    assert torch.equal(torch.ops.omnitok.vq_argmin(z, E), ops.vq_argmin(z, E))
    x, w, b = dev(rnd(300, 512, seed=1)), dev(rnd(192, 512, seed=2) * 0.05), dev(rnd(192, seed=3))
    assert torch.equal(torch.ops.omnitok.linear(x, w, b), ops.linear(x, w, b))
    gm = dev(rnd(512, seed=4) + 1.0)
    assert torch.equal(torch.ops.omnitok.layernorm(x, gm, None), ops.layernorm(x, gm, None))


def test_token_resample(ops):
    """pooling blocks / deferred pools vs the torch ops the reference calls (bit-exact: adds and
    power-of-two scalings in the same order)."""
    x = rnd(3, 8, 8, 128, seed=96)  # n gh gw D
    nchw = x.permute(0, 3, 1, 2)
    assert torch.equal(ops.token_resample(dev(x), "avg2d").cpu(), F.avg_pool2d(nchw, 2).permute(0, 2, 3, 1))
    assert torch.equal(ops.token_resample(dev(x), "max2d").cpu(), F.max_pool2d(nchw, 2).permute(0, 2, 3, 1))
    up = F.interpolate(nchw, scale_factor=2, mode="nearest").permute(0, 2, 3, 1)
    assert torch.equal(ops.token_resample(dev(x), "up2d").cpu(), up)
    for T in (1, 2, 4, 5):
        v = rnd(2, T, 12, 64, seed=97 + T)  # B T S D
        y = v.permute(0, 3, 1, 2).unsqueeze(-1)  # b d t s 1
        want = y[:, :, :1]
        if T > 1 and (T - 1) // 2 > 0:
            want = torch.cat([want, F.avg_pool3d(y[:, :, 1:], (2, 1, 1))], dim=2)
        got = ops.token_resample(dev(v), "avg_t").cpu()
        assert torch.equal(got, want.squeeze(-1).permute(0, 2, 3, 1)), T
        want = y[:, :, :1]
        if T > 1:
            want = torch.cat([want, F.interpolate(y[:, :, 1:], scale_factor=(2, 1, 1), mode="nearest")], dim=2)
        got = ops.token_resample(dev(v), "up_t").cpu()
        assert torch.equal(got, want.squeeze(-1).permute(0, 2, 3, 1)), T


def test_patchify_raw_and_padded(ops):
    v = rnd(2, 3, 5, 16, 16, seed=98)
    raw = ops.patchify_ln(dev(v), 1, 2, 2, 4)  # no LayerNorm: the Conv3d(kernel == stride) im2col rows
    assert torch.equal(raw.cpu(), orc.patchify(v[:, :, 1:], 4, 2).reshape(-1, 96))
    g, b = rnd(48, seed=99) + 1.0, rnd(48, seed=100)
    pad = ops.patchify_ln(dev(v), 0, 1, 1, 4, dev(g), dev(b), ldo=64)
    want = F.layer_norm(orc.patchify(v[:, :, :1], 4, 1).reshape(-1, 48), (48,), g, b, 1e-5)
    assert maxerr(pad[:, :48], want) < 1e-5 and (pad[:, 48:] == 0).all()


def test_vae_sample_and_post_vq(ops):
    """--use_vae kernels against the oracle's DiagonalGaussianDistribution restatement."""
    c = GoldenCase("vae_s2_sdpa_r64_vid")
    B, thw = 3, 200
    x = rnd(B, thw, 512, seed=93)
    x[0, :4] *= 40.0  # drive some logvars into the [-30, 20] clamp
    w, b = c.sd["pre_vq_conv.1.weight"].clone(), c.sd["pre_vq_conv.1.bias"]
    w[8:] *= 6.0
    noise = rnd(B, 8, thw, seed=94)
    h = F.linear(x, w, b).permute(0, 2, 1)  # b 2c thw
    mean, logvar = torch.chunk(h, 2, dim=1)
    clamped = torch.clamp(logvar, -30.0, 20.0)
    assert (clamped != logvar).any(), "test input does not reach the clamp"
    ref = mean + torch.exp(0.5 * clamped) * noise
    z, mom = ops.vae_sample(dev(x), dev(w), dev(b), dev(noise), return_moments=True)
    assert maxerr(mom, h) < 2e-4 * max(1.0, h.abs().max().item() / 10)
    # the sample arithmetic alone, on the kernel's own moments (the inflated logvars amplify the
    # fp32 summation-order noise of h by 0.5*std, so the end-to-end bound is looser)
    dmean, dlogvar = torch.chunk(mom.cpu(), 2, dim=1)
    ref_dev = dmean + torch.exp(0.5 * torch.clamp(dlogvar, -30.0, 20.0)) * noise
    rel = ((z.cpu() - ref_dev).abs() / ref_dev.abs().clamp_min(1.0)).max().item()
    assert rel < 2e-6, rel
    rel = ((z.cpu() - ref).abs() / ref.abs().clamp_min(1.0)).max().item()
    assert rel < 2e-4, rel
    mode = ops.vae_sample(dev(x), dev(w), dev(b), None)
    assert maxerr(mode, mean) < 2e-4
    # post_vq on continuous latents, both layouts
    pw, pb = c.sd["post_vq_conv.1.weight"], c.sd["post_vq_conv.1.bias"]
    zl = rnd(B, thw, 8, seed=95)
    want = F.linear(zl, pw, pb)
    assert maxerr(ops.post_vq(dev(zl), dev(pw), dev(pb), channel_first=False), want) < 1e-5
    assert maxerr(ops.post_vq(dev(zl.permute(0, 2, 1).contiguous()), dev(pw), dev(pb), channel_first=True),
                  want) < 1e-5


VQ_VARIANTS = pytest.mark.parametrize("variant", [0, 1, 2], ids=["valu_epilogue", "fifth_mfma_step", "lds_codebook"])


@VQ_VARIANTS
@pytest.mark.parametrize("n_codes", [8192, 16384])
def test_vq_argmin_bit_exact_vs_reference_kat(ops, n_codes, variant):
    """ids bit-exact against the reference's own Codebook.forward outputs (golden KAT), including
    duplicated code rows (ties -> lowest index) and scaled / exact-code inputs -- for every arm of the search
    kernel (csrc/vq.hip "vq_variant": the default, the fifth-MFMA-step chain, the LDS-staged codebook)."""
    from omnitokenizer_amd import _lib
    g = np.load(os.path.join(GOLDEN, f"vq_kat_{n_codes}.npz"))
    z, E, ids_ref = torch.from_numpy(g["z"]), torch.from_numpy(g["codebook"]), g["ids"].astype(np.int64)
    _lib.set_option("vq_variant", variant)
    try:
        ids = ops.vq_argmin(dev(z), dev(E)).cpu().numpy()
        assert np.array_equal(ids, ids_ref), f"{(ids != ids_ref).sum()} of {ids.size} ids differ"
        assert (ids[4096:4160] == 17).all() and (ids[4160:4200] == 3).all()
        # ragged sizes (not a multiple of the 256-row workgroup) and the empty input
        for n in (1, 31, 257, 1000):
            assert np.array_equal(ops.vq_argmin(dev(z[:n]), dev(E)).cpu().numpy(), ids_ref[:n])
        assert ops.vq_argmin(dev(z[:0]), dev(E)).numel() == 0
        # un-normalised inputs (|xx - dot| far from the unit sphere's range) against the C oracle
        rng = np.random.default_rng(11)
        zz = (rng.standard_normal((3000, 8)) * 3.0).astype(np.float32)
        EE = rng.standard_normal((2048, 8)).astype(np.float32)
        got = ops.vq_argmin(dev(torch.from_numpy(zz)), dev(torch.from_numpy(EE))).cpu().numpy()
        assert np.array_equal(got, c_oracle.vq_argmin(zz, EE))
    finally:
        _lib.set_option("vq_variant", 1)  # the default


@VQ_VARIANTS
@pytest.mark.parametrize("split", [1, 2, 4, 16, 3])
def test_vq_argmin_code_range_splits(ops, split, variant):
    """The load-balancing code-range split (64-bit atomicMin merge of distance key | index) returns
    the same first-minimum ids whatever the number of splits, ties included."""
    from omnitokenizer_amd import _lib
    g = np.load(os.path.join(GOLDEN, "vq_kat_8192.npz"))
    z, E, ids_ref = torch.from_numpy(g["z"]), torch.from_numpy(g["codebook"]), g["ids"].astype(np.int64)
    _lib.set_option("vq_split", split)
    _lib.set_option("vq_variant", variant)
    try:
        for n in (6144, 1000, 31):
            ids = ops.vq_argmin(dev(z[:n]), dev(E)).cpu().numpy()
            assert np.array_equal(ids, ids_ref[:n]), f"split {split}, n {n}: {(ids != ids_ref[:n]).sum()} differ"
        # exact duplicates far apart in the codebook land in different splits: lowest index wins
        E2 = E.clone()
        E2[8000] = E2[100]
        zz = E2[[100, 8000, 100]].contiguous()
        assert ops.vq_argmin(dev(zz), dev(E2)).cpu().tolist() == [100, 100, 100]
    finally:
        _lib.set_option("vq_split", 0)
        _lib.set_option("vq_variant", 1)


@pytest.mark.parametrize("split", [0, 1, 4])
def test_vq_argmax_cos_bit_exact_vs_reference_kat(ops, split):
    """External cosine-similarity codebook: ids bit-exact against the reference's CosineSimCodebook.forward."""
    from omnitokenizer_amd import _lib
    g = np.load(os.path.join(GOLDEN, "vq_cos_kat_8192.npz"))
    z, E, ids_ref = torch.from_numpy(g["z"]), torch.from_numpy(g["codebook"]), g["ids"].astype(np.int64)
    _lib.set_option("vq_split", split)
    try:
        for n in (4096, 1000, 31):
            ids = ops.vq_argmax_cos(dev(z[:n]), dev(E)).cpu().numpy()
            assert np.array_equal(ids, ids_ref[:n]), f"{(ids != ids_ref[:n]).sum()} of {n} ids differ"
    finally:
        _lib.set_option("vq_split", 0)
    assert np.array_equal(c_oracle.vq_argmax_cos(g["z"][:512], g["codebook"]), ids_ref[:512])


@pytest.mark.parametrize("split", [0, 1, 4])
def test_vq_argmin_cdist_bit_exact_vs_reference_kat(ops, split):
    """External Euclidean codebook: ids bit-exact against the reference's EuclideanCodebook.forward (the
    correctly rounded sqrt of the clamped squared distance takes part in the ordering)."""
    from omnitokenizer_amd import _lib
    g = np.load(os.path.join(GOLDEN, "vq_cdist_kat_8192.npz"))
    z, E, ids_ref = torch.from_numpy(g["z"]), torch.from_numpy(g["codebook"]), g["ids"].astype(np.int64)
    _lib.set_option("vq_split", split)
    try:
        for n in (4096, 1000, 31):
            ids = ops.vq_argmin_cdist(dev(z[:n]), dev(E)).cpu().numpy()
            assert np.array_equal(ids, ids_ref[:n]), f"{(ids != ids_ref[:n]).sum()} of {n} ids differ"
    finally:
        _lib.set_option("vq_split", 0)
    rng = np.random.default_rng(7)
    zz = rng.standard_normal((20000, 8), dtype=np.float32)
    EE = rng.standard_normal((1024, 8), dtype=np.float32)
    ids = ops.vq_argmin_cdist(dev(torch.from_numpy(zz)), dev(torch.from_numpy(EE))).cpu().numpy()
    assert np.array_equal(ids[:4096], c_oracle.vq_argmin_cdist(zz[:4096], EE))


def test_vq_argmin_large_random_vs_c_oracle(ops):
    rng = np.random.default_rng(5)
    E = rng.standard_normal((8192, 8), dtype=np.float32)
    z = rng.standard_normal((65536, 8), dtype=np.float32)
    z /= np.linalg.norm(z, axis=1, keepdims=True)
    ids = ops.vq_argmin(dev(torch.from_numpy(z)), dev(torch.from_numpy(E))).cpu().numpy()
    sub = slice(0, 8192)  # the scalar C oracle is slow: check a slice exactly ...
    assert np.array_equal(ids[sub], c_oracle.vq_argmin(z[sub], E))
    # ... and all rows through an optimality property in fp64
    zz, Ed = z.astype(np.float64), E.astype(np.float64)
    chosen = ((zz - Ed[ids]) ** 2).sum(1)
    # cheap global check: chosen distance never exceeds the distance to 64 random other codes
    other = rng.integers(0, 8192, (65536, 64))
    dist_other = ((zz[:, None, :] - Ed[other]) ** 2).sum(-1).min(1)
    assert (chosen <= dist_other + 1e-6).all()


@pytest.mark.parametrize("split", [0, 1, 2, 4, 16, 3])
def test_vq_screened_search_is_bit_identical(ops, split):
    """The fp16-screened nearest-code search (omnitok_vq_argmin_screened: one fp16 MFMA per tile with a rigorous error
    bound, exact re-evaluation of the candidates) returns the ids of the exact sweep and of the reference, bit for bit:
    the reference's KATs (with duplicated codes: lowest index), every code-range split, ragged / empty sizes, inputs far
    off the unit sphere, codebooks full of duplicates (more candidate tiles than the list holds), inputs outside fp16's
    range and non-finite rows (those must match whatever the exact kernel returns)."""
    from omnitokenizer_amd import _lib
    _lib.set_option("vq_screen_split", split)
    try:
        for n_codes in (8192, 16384):
            g = np.load(os.path.join(GOLDEN, f"vq_kat_{n_codes}.npz"))
            z, E, ids_ref = torch.from_numpy(g["z"]), torch.from_numpy(g["codebook"]), g["ids"].astype(np.int64)
            ids = ops.vq_argmin_screened(dev(z), dev(E)).cpu().numpy()
            assert np.array_equal(ids, ids_ref), f"{n_codes}: {(ids != ids_ref).sum()} of {ids.size} ids differ"
            for n in (1, 31, 257, 1000):
                assert np.array_equal(ops.vq_argmin_screened(dev(z[:n]), dev(E)).cpu().numpy(), ids_ref[:n])
            assert ops.vq_argmin_screened(dev(z[:0]), dev(E)).numel() == 0
        rng = np.random.default_rng(11)
        zz = (rng.standard_normal((3000, 8)) * 3.0).astype(np.float32)
        EE = rng.standard_normal((2048, 8)).astype(np.float32)
        got = ops.vq_argmin_screened(dev(torch.from_numpy(zz)), dev(torch.from_numpy(EE))).cpu().numpy()
        assert np.array_equal(got, c_oracle.vq_argmin(zz, EE))
        # duplicates: two far apart, and a codebook where ONE vector fills 9 tiles (more candidates than the list holds)
        g = np.load(os.path.join(GOLDEN, "vq_kat_8192.npz"))
        E2 = torch.from_numpy(g["codebook"]).clone()
        E2[8000] = E2[100]
        q = E2[[100, 8000, 100]].contiguous()
        assert ops.vq_argmin_screened(dev(q), dev(E2)).cpu().tolist() == [100, 100, 100]
        E3 = E2.clone()
        E3[3000:3000 + 9 * 32] = E3[77]
        q3 = torch.cat([E3[[77, 3100]], torch.from_numpy(g["z"][:500])])
        want = ops.vq_argmin(dev(q3), dev(E3)).cpu()
        assert torch.equal(ops.vq_argmin_screened(dev(q3), dev(E3)).cpu(), want) and want[:2].tolist() == [77, 77]
        # rows outside fp16's range, tiny rows, zeros, inf and NaN rows: whatever the exact kernel answers
        zb = torch.from_numpy(g["z"][:256]).clone()
        zb[0] *= 1e6
        zb[1] *= 4e4
        zb[2] *= 1e-30
        zb[3] = 0.0
        zb[4, 3] = float("inf")
        zb[5, 0] = float("nan")
        zb[6] = float("-inf")
        E1 = torch.from_numpy(g["codebook"])
        assert torch.equal(ops.vq_argmin_screened(dev(zb), dev(E1)).cpu(), ops.vq_argmin(dev(zb), dev(E1)).cpu())
        # a codebook outside fp16's range / with a non-finite entry: every row takes the exact evaluation
        for bad in (7e4, float("inf"), float("nan")):
            Eb = E1.clone()
            Eb[5000, 2] = bad
            zq = torch.from_numpy(g["z"][:300])
            assert torch.equal(ops.vq_argmin_screened(dev(zq), dev(Eb)).cpu(), ops.vq_argmin(dev(zq), dev(Eb)).cpu())
    finally:
        _lib.set_option("vq_screen_split", 0)


def test_vq_screened_large_random_vs_exact(ops):
    """262 144 unit-norm rows (the C3 batch is 163 840) x 8192 and 16 384 codes, and heavy-tailed codebooks: screened ==
    exact on every row; a slice against the scalar C oracle."""
    rng = np.random.default_rng(5)
    for n_codes, scale in ((8192, 1.0), (16384, 1.0), (8192, 0.05), (8192, 30.0)):
        E = (rng.standard_normal((n_codes, 8)) * scale).astype(np.float32)
        if scale == 30.0:
            E *= rng.standard_t(3, size=(n_codes, 1)).astype(np.float32)   # a few codes with very large norms
        z = rng.standard_normal((262144, 8), dtype=np.float32)
        z /= np.linalg.norm(z, axis=1, keepdims=True)
        zd, Ed = dev(torch.from_numpy(z)), dev(torch.from_numpy(E))
        a = ops.vq_argmin_screened(zd, Ed).cpu().numpy()
        b = ops.vq_argmin(zd, Ed).cpu().numpy()
        assert np.array_equal(a, b), f"{n_codes} codes x {scale}: {(a != b).sum()} rows differ"
        assert np.array_equal(a[:2048], c_oracle.vq_argmin(z[:2048], E))


def test_vq_idempotence_full_codebook(ops):
    """Quantising the code vectors themselves returns their own indices (property test at the
    full codebook size)."""
    c = GoldenCase("s2_sdpa_r64_img")
    E = c.sd["codebook.embeddings"]
    ids = ops.vq_argmin(dev(E), dev(E)).cpu()
    assert torch.equal(ids, torch.arange(E.shape[0]))
    assert torch.equal(ops.vq_argmin_screened(dev(E), dev(E)).cpu(), ids)
