"""GPU (-m gpu): the fp16-split spatial attention (csrc/attn_h2.hip: omnitok_attn_pack + omnitok_attn_spatial_h2)
against an fp64 restatement of reference attention.py:417-483, beside the fp32-MFMA kernel it replaces on the
engine path.  Tolerances: the same 1e-5 / 2e-5 absolute bars as tests/test_gpu_ops.py::test_attn_spatial*, and the
packed operands must reproduce the fp32 q / k / v to 2^-21 relative (11 + 11 significand bits)."""
import pytest
import torch

from oracle import omnitok_oracle as orc
from tests.helpers import GoldenCase

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "needs the MI355X"
    from omnitokenizer_amd import ops as _ops
    return _ops


@pytest.fixture(params=[6, 3, 1, 0, 2, 4, 5, 7], ids=["wide_deferred_rescale", "hoisted_32key", "pipelined_dma", "reg_staged", "hoisted_64key", "wide_interleaved", "wide", "wide_8waves"])
def variant(request):
    """every build of the attention kernel ("attn_h2_variant": 1 = software-pipelined + LDS-DMA, 0 = register-staged, 2 / 3 = fragment
    reads hoisted in front of the MFMA chains with 64- / 32-key tiles)."""
    from omnitokenizer_amd import _lib
    _lib.set_option("attn_h2_variant", request.param)
    yield request.param
    _lib.set_option("attn_h2_variant", 6)


def dev(t):
    return t.contiguous().cuda()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def maxerr(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item()


def ref_attention(q, k, v, bias=None):
    """q, k, v [Bn, N, h, d] (already prepared) -> [Bn*N, h*d] in fp64."""
    s = torch.einsum("bihd,bjhd->bhij", q.double(), k.double())
    if bias is not None:
        s = s + bias.double()
    o = torch.einsum("bhij,bjhd->bihd", s.softmax(-1), v.double())
    return o.reshape(q.shape[0] * q.shape[1], -1)


def prepared(q, k, qs, ks, N, rope):
    """fp32 restatement of reference attention.py:417-437 on [Bn, N, h, d] tensors."""
    if rope:
        cos, sin = orc.rope_table(N, q.shape[-1])
        q, k = orc.apply_rope(q, cos, sin), orc.apply_rope(k, cos, sin)
    return orc.l2norm(q) * qs * 8.0, orc.l2norm(k) * ks


def unpack_qk(plane_words, Bn, N, h, scale):
    """[Bn*N*h*64] int32 words in the Q / K block layout of attn_h2.hip -> fp32 [Bn*N, h*64] (hi + lo) / scale."""
    x = plane_words.view(torch.float16).float().reshape(Bn, h, N // 32, 2, 4, 2, 32, 8)  # seq head blk pl ks h tok e
    x = x[:, :, :, 0] + x[:, :, :, 1]
    x = x.permute(0, 2, 5, 1, 3, 4, 6)  # seq blk tok head ks h e
    return x.reshape(Bn * N, h * 64) / scale


def unpack_v(plane_words, Bn, N, h, scale):
    x = plane_words.view(torch.float16).float().reshape(Bn, h, N // 32, 2, 2, 2, 2, 32, 2, 4)  # .. pl j mt hh d ii r
    x = x[:, :, :, 0] + x[:, :, :, 1]                                                          # seq head blk j mt hh d ii r
    x = x.permute(0, 2, 3, 7, 5, 8, 1, 4, 6)   # seq blk j ii hh r | head mt d  (key = 16 j + 8 ii + 4 hh + r)
    return x.reshape(Bn * N, h * 64) / scale


def pow2_scale(bound):
    import math
    m, x = math.frexp(bound)
    return 2.0 ** -(x - 15)


@pytest.mark.parametrize("rope", [True, False])
def test_attn_pack_layout_and_split(ops, rope):
    Bn, N, h, d = 2, 128, 8, 64
    q, kv = rnd(Bn * N, h * d, seed=151), rnd(Bn * N, 2 * h * d, seed=152, scale=3.0)
    qs, ks = rnd(d, seed=153) * 0.1 + 1, rnd(d, seed=154) * 0.1 + 1
    cos = sin = None
    if rope:
        cos, sin = orc.rope_table(N, d)
    qd, kvd = dev(q), dev(kv)
    packed, bounds = ops.attn_pack(qd, kvd[:, : h * d], kvd[:, h * d:], N, h, dev(qs), dev(ks),
                                   None if cos is None else dev(cos), None if sin is None else dev(sin))
    # the fp32 values the pack kernel splits are those of qk_prep (same arithmetic)
    q2, kv2 = dev(q), dev(kv)
    ops.qk_prep_(q2, kv2[:, : h * d], N, h, dev(qs), dev(ks), None if cos is None else dev(cos),
                 None if sin is None else dev(sin))
    pk = packed.cpu()
    for words, ref, bound in ((pk[0], q2.cpu(), bounds[0]), (pk[1], kv2[:, : h * d].cpu(), bounds[1])):
        got = unpack_qk(words, Bn, N, h, pow2_scale(bound))
        assert ((got - ref).abs() <= ref.abs() * 2.0 ** -21 + 1e-9).all()
    got = unpack_v(pk[2], Bn, N, h, pow2_scale(bounds[2]))
    ref = kv[:, h * d:]
    assert ((got - ref).abs() <= ref.abs() * 2.0 ** -21 + 1e-9).all()


@pytest.mark.parametrize("rope", [True, False])
@pytest.mark.parametrize("Bn,N", [(3, 64), (2, 192), (1, 1024), (2, 576)])
def test_attn_h2_vs_fp64(ops, variant, Bn, N, rope):
    h, d = 8, 64
    q, k, v = rnd(Bn, N, h, d, seed=161), rnd(Bn, N, h, d, seed=162), rnd(Bn, N, h, d, seed=163, scale=2.0)
    qs, ks = rnd(d, seed=164) * 0.1 + 1, rnd(d, seed=165) * 0.1 + 1
    qp, kp = prepared(q, k, qs, ks, N, rope)
    ref = ref_attention(qp, kp, v)
    cos = sin = None
    if rope:
        cos, sin = orc.rope_table(N, d)
        cos, sin = dev(cos), dev(sin)
    qd = dev(q.reshape(Bn * N, h * d))
    kvd = dev(torch.cat([k.reshape(Bn * N, h * d), v.reshape(Bn * N, h * d)], dim=1))
    packed, bounds = ops.attn_pack(qd, kvd[:, : h * d], kvd[:, h * d:], N, h, dev(qs), dev(ks), cos, sin)
    out = ops.attn_spatial_h2(packed, bounds, Bn, N, h)
    # the fp32-MFMA kernel on the same operands
    ops.qk_prep_(qd, kvd[:, : h * d], N, h, dev(qs), dev(ks), cos, sin)
    out32 = ops.attn_spatial(qd, kvd[:, : h * d], kvd[:, h * d:], Bn, N, h)
    e_h2, e_32 = maxerr(out, ref), maxerr(out32, ref)
    print(f"attn variant={variant} Bn={Bn} N={N} rope={rope}: fp16-split err {e_h2:.2e}, fp32-MFMA err {e_32:.2e}")
    assert e_h2 < 1e-5
    assert e_h2 < 4 * e_32 + 2e-6


def test_attn_h2_forced_rescale(ops, variant):
    """A key whose logit towers over the others late in the sweep forces the online-softmax rescale branch."""
    Bn, N, h, d = 1, 256, 8, 64
    q = orc.l2norm(rnd(Bn, N, h, d, seed=64)) * 8.0
    k = orc.l2norm(rnd(Bn, N, h, d, seed=65))
    k[0, 200] = q[0, 17] / 8.0  # logit 8 for query 17 at key 200 (the largest a unit-norm pair can reach)
    v = rnd(Bn, N, h, d, seed=66)
    ones = torch.ones(d)
    ref = ref_attention(q, k, v)
    # identity scales and no RoPE: the pack kernel re-normalises already unit vectors (error ~1 ulp)
    qd = dev((q / 8.0).reshape(N, h * d))
    kvd = dev(torch.cat([k.reshape(N, h * d), v.reshape(N, h * d)], dim=1))
    packed, bounds = ops.attn_pack(qd, kvd[:, : h * d], kvd[:, h * d:], N, h, dev(ones), dev(ones))
    out = ops.attn_spatial_h2(packed, bounds, Bn, N, h)
    assert torch.isfinite(out).all()
    assert maxerr(out, ref) < 2e-5


def test_attn_h2_legacy_bias(ops, variant):
    c = GoldenCase("s1_legacy_r64_img")
    p = "encoder.enc_spatial_transformer.layers.0.1.spatial_rel_pos_bias"
    gh = gw = 8
    Bn, N, h, d = 2, 64, 8, 64
    full = orc.continuous_position_bias(c.sd, p, gh, gw)                  # h, N, N
    tab = orc.continuous_position_bias_table(c.sd, p, gh, gw)             # h, 2gh-1, 2gw-1
    tab_dev = dev(tab.permute(1, 2, 0).reshape(-1, h))                    # [(2gh-1)(2gw-1), h]
    q, k, v = rnd(Bn, N, h, d, seed=67), rnd(Bn, N, h, d, seed=68), rnd(Bn, N, h, d, seed=69)
    qs, ks = rnd(d, seed=70) * 0.1 + 1, rnd(d, seed=71) * 0.1 + 1
    qp, kp = prepared(q, k, qs, ks, N, False)
    ref = ref_attention(qp, kp, v, bias=full[None])
    qd = dev(q.reshape(Bn * N, h * d))
    kvd = dev(torch.cat([k.reshape(Bn * N, h * d), v.reshape(Bn * N, h * d)], dim=1))
    packed, bounds = ops.attn_pack(qd, kvd[:, : h * d], kvd[:, h * d:], N, h, dev(qs), dev(ks))
    out = ops.attn_spatial_h2(packed, bounds, Bn, N, h, tab_dev, gh, gw)
    assert maxerr(out, ref) < 2e-5


@pytest.mark.parametrize("gh,gw", [(16, 16), (32, 32), (8, 16), (16, 8), (16, 12), (64, 32)])
def test_attn_h2_legacy_bias_grids(ops, variant, gh, gw):
    """r06: the 64-query kernels (variants 4 - 7) read the legacy relative-position bias (reference attention.py:453-483, 535-583) from
    a per-head table staged in LDS -- power-of-two grid widths >= 8 with tables up to 32 KiB; other grids (16 x 12: width not a power
    of two; 64 x 32: 127 x 63 floats > 32 KiB) keep the global gather of variant 3.  Every variant and grid against fp64 with the
    dense bias of the oracle's ContinuousPositionBias."""
    c = GoldenCase("s1_legacy_r64_img")
    p = "encoder.enc_spatial_transformer.layers.0.1.spatial_rel_pos_bias"
    N = gh * gw
    Bn, h, d = (1 if N > 1024 else 2), 8, 64
    full = orc.continuous_position_bias(c.sd, p, gh, gw)                  # h, N, N
    tab = orc.continuous_position_bias_table(c.sd, p, gh, gw)             # h, 2gh-1, 2gw-1
    tab_dev = dev(tab.permute(1, 2, 0).reshape(-1, h))
    q, k, v = rnd(Bn, N, h, d, seed=167), rnd(Bn, N, h, d, seed=168), rnd(Bn, N, h, d, seed=169)
    qs, ks = rnd(d, seed=170) * 0.1 + 1, rnd(d, seed=171) * 0.1 + 1
    qp, kp = prepared(q, k, qs, ks, N, False)
    ref = ref_attention(qp, kp, v, bias=full[None])
    qd = dev(q.reshape(Bn * N, h * d))
    kd, vd = dev(k.reshape(Bn * N, h * d)), dev(v.reshape(Bn * N, h * d))
    packed, bounds = ops.attn_pack(qd, kd, vd, N, h, dev(qs), dev(ks))
    out = ops.attn_spatial_h2(packed, bounds, Bn, N, h, tab_dev, gh, gw)
    assert torch.isfinite(out).all()
    assert maxerr(out, ref) < 2e-5, (variant, gh, gw, maxerr(out, ref))
    planes, scales = ops.attn_spatial_h2_planes(packed, bounds, Bn, N, h, tab_dev, gh, gw)   # the engine's form: planes out
    got = ops.pl_unpack_planes(planes, Bn * N, h * d) * scales.double()[:, None]
    assert maxerr(got, out) <= bounds[2] * 2.0 ** -21


def test_attn_h2_per_clip_ranges_and_batch_independence(ops, variant):
    """v of very different magnitude per clip with device-side per-clip bounds: every clip keeps its relative
    accuracy, and a clip's result does not depend on what else is in the batch (bitwise)."""
    clips, T, N, h, d = 3, 2, 64, 8, 64
    Bn = clips * T
    q, k = rnd(Bn, N, h, d, seed=171), rnd(Bn, N, h, d, seed=172)
    amp = torch.tensor([1e-3, 1.0, 3e3]).repeat_interleave(T).reshape(Bn, 1, 1, 1)
    v = rnd(Bn, N, h, d, seed=173) * amp
    ones = torch.ones(d)
    qp, kp = prepared(q, k, ones, ones, N, False)
    ref = ref_attention(qp, kp, v).reshape(clips, -1)
    slots = torch.zeros(clips, 2)
    slots[:, 1] = v.reshape(clips, -1).abs().amax(dim=1)   # slot [2 c + 1] like the engine's row statistics
    slots_d = dev(slots.reshape(-1))

    def run(sel):
        qd = dev(q[sel].reshape(-1, h * d))
        kvd = dev(torch.cat([k[sel].reshape(-1, h * d), v[sel].reshape(-1, h * d)], dim=1))
        sl = slots_d.reshape(clips, 2)[sel[::T] // T].reshape(-1).contiguous()
        n = len(sel)
        packed, bounds = ops.attn_pack(qd, kvd[:, : h * d], kvd[:, h * d:], N, h, dev(ones), dev(ones), v_bound=1.01,
                                       v_bound_dev=sl[1:], v_bound_stride=2, rows_per_clip=T * N)
        return ops.attn_spatial_h2(packed, bounds, n, N, h, v_bound_dev=sl[1:], v_bound_stride=2, seq_per_clip=T)

    full = run(torch.arange(Bn)).cpu().reshape(clips, -1)
    for c in range(clips):
        rel = (full[c].double() - ref[c]).abs().max().item() / float(amp.reshape(clips, T)[c, 0])
        assert rel < 1e-5, (c, rel)
    alone = run(torch.arange(2 * T, 3 * T)).cpu().reshape(-1)
    assert torch.equal(alone, full[2])


@pytest.mark.parametrize("mode", [0, 1], ids=["fp32_mfma", "fp16x2"])
@pytest.mark.parametrize("name", ["s2_sdpa_r64_vid", "s1_legacy_r64_img", "s2_sdpa_r128_vid_16k", "s2_sdpa_r256_img",
                                  "var_up_n_r64_vid"])
def test_engine_attn_modes_vs_golden(mode, name):
    """encode ids bit-exact and decode pixels within 1e-4 of the reference in both attention modes of the engine."""
    from omnitokenizer_amd import OmniTokenizer_VQGAN, _lib
    c = GoldenCase(name)
    _lib.set_option("attn_mode", mode)
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
        _lib.set_option("attn_mode", 1)


@pytest.mark.parametrize("M,N_tok,rpc", [(256, 64, 0), (1024, 256, 256), (49152, 1024, 12288), (640, 64, 128)])
def test_qkv_gemm_writes_v_planes(ops, M, N_tok, rpc):
    """omnitok_gemm_h2_vpack: the V columns of the merged to_q | to_kv launch land in the packed fp16 planes bit for bit
    as if they had been stored as fp32 and packed by omnitok_attn_pack; the q | k columns are those of the plain launch."""
    D, h = 256, 4
    x = rnd(M, D, seed=171, scale=2.0)
    w = rnd(3 * D, D, seed=172, scale=D ** -0.5)
    gam, bet = rnd(D, seed=173) * 0.1 + 1, rnd(D, seed=174) * 0.1
    xd, pk = dev(x), ops.h2_pack_weight(dev(w))
    lnb = float(D ** 0.5 * gam.abs().max() + bet.abs().max())
    kw, vkw, vdev = {}, {}, None
    if rpc:
        bounds = torch.zeros(M // rpc, 2, device="cuda")
        st = ops.row_stats(xd, bounds=bounds, rows_per_clip=rpc)
        vdev = bounds.view(-1)[1:]  # slot 1 of every clip: max ||x_row||
        kw = dict(a_bound_dev=bounds, a_bound_stride=2, rows_per_clip=rpc)
        vkw = dict(v_bound_dev=vdev, v_bound_stride=2)
        a_bound, v_bound = 1.01, 1.01 * float(w[2 * D:].norm(dim=1).max())
    else:
        st = ops.row_stats(xd)
        a_bound, v_bound = float(x.abs().max()), 1.01 * float((x @ w[2 * D:].T).abs().max())
    plain = ops.linear_h2(xd, pk, a_bound, ln=(st, dev(gam), dev(bet)), ln_cols=D, ln_bound=lnb, **kw)
    qk, vp = ops.linear_h2_vpack(xd, pk, a_bound, (st, dev(gam), dev(bet)), D, lnb, 2 * D, N_tok, h, v_bound, **kw, **vkw)
    assert torch.equal(qk, plain[:, : 2 * D])
    qs = torch.ones(64, device="cuda")
    v32 = plain[:, 2 * D:]
    packed, _ = ops.attn_pack(plain[:, :D], plain[:, D: 2 * D], v32, N_tok, h, qs, qs, v_bound=v_bound,
                              v_bound_dev=vdev, v_bound_stride=2, rows_per_clip=rpc)
    assert torch.equal(vp, packed[2])
