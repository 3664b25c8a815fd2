"""GPU (-m gpu): OmniTokenizer_VQGAN.encode()/decode() on the MI355X against the committed outputs
of the reference (tests/golden) and the CPU oracle, plus size-independent properties at the
BASELINE.json batch sizes.

Parity tiers (SURVEY.md sections 7, 8(c)):
  (i)   VQ kernel on the reference's own z: ids bit-exact;
  (ii)  decode(reference ids): pixels within 1e-4 abs of the reference;
  (iii) end-to-end encode: ids equal to the reference except at provable near-ties (z differs from
        the reference's by fp32 summation order only; a flip is accepted only if the two candidate
        codes' fp64 distances differ by < 1e-5 relative) -- expected count 0.
"""
import os

import numpy as np
import pytest
import torch

from oracle import omnitok_oracle as orc
from tests.helpers import (E2E_CASES, EXT_CASES, FULL_CASES, GOLDEN, HEAVY_BATCH_CASE, HEAVY_CASES, VAE_CASES, VARIANT_CASES,
                           GoldenCase)

pytestmark = pytest.mark.gpu

PIXEL_TOL = 1e-4
Z_TOL = 2e-5


@pytest.fixture(scope="module")
def models():
    cache = {}

    def get(case):
        key = (case.stage, case.mode, tuple(sorted(case.overrides.items())), case.profile)
        if key not in cache:
            from omnitokenizer_amd import OmniTokenizer_VQGAN
            m = OmniTokenizer_VQGAN(case.args, attention_mode=case.mode)
            missing = m.load_state_dict(case.sd, strict=True)
            assert not missing.missing_keys and not missing.unexpected_keys
            cache[key] = m.cuda().eval()
        return cache[key]
    return get


def assert_ids_match_or_near_tie(ids, ids_ref, z_ours, codebook, name):
    ids, ids_ref = ids.reshape(-1).cpu(), ids_ref.reshape(-1)
    bad = (ids != ids_ref).nonzero().flatten()
    if bad.numel() == 0:
        return 0
    # allowance: the observed order of magnitude (2-3 of 163 840 on the full bench batch = 1.8e-5; VERDICT r05: N // 50000), and
    # every one of them must be a provable near-tie below
    assert bad.numel() <= max(1, ids.numel() // 50000), f"{name}: {bad.numel()} of {ids.numel()} ids differ"
    z = z_ours.reshape(-1, z_ours.shape[-1]).cpu().double()[bad]
    E = codebook.double()
    d_ours = ((z - E[ids[bad]]) ** 2).sum(1)
    d_ref = ((z - E[ids_ref[bad]]) ** 2).sum(1)
    gap = (d_ours - d_ref).abs() / d_ref.clamp_min(1e-12)
    assert (gap < 1e-5).all(), f"{name}: id flips that are not near-ties, gaps {gap.tolist()}"
    return int(bad.numel())


@pytest.mark.parametrize("name", E2E_CASES + FULL_CASES + VARIANT_CASES)
def test_encode_decode_vs_reference_golden(models, name):
    c = GoldenCase(name)
    m = models(c)
    from omnitokenizer_amd import ops
    x = c.x.cuda()
    ids, z = m.encode(x, c.is_image, return_latents=True)
    assert ids.dtype == torch.int64 and tuple(ids.shape) == tuple(c.ids.shape)
    # the shape functions behind the operators' fake implementations agree with the engine
    F_, H_, W_ = (1 if c.is_image else x.shape[2]), x.shape[-2], x.shape[-1]
    assert m.latent_dims(F_, H_, W_) == tuple(ids.shape[1:]) == m._shape("encode", F_, H_, W_)
    assert m.pixel_dims(*ids.shape[1:]) == m._shape("decode", *ids.shape[1:])
    # tier (i): the quantiser alone on the reference's z -> bit-exact
    ids_on_ref_z = ops.vq_argmin(c.z.cuda(), m.codebook.embeddings.data)
    assert torch.equal(ids_on_ref_z.cpu(), c.ids), "VQ kernel not bit-exact on the reference's z"
    # tier (iii)
    zerr = (z.cpu() - c.z).abs().max().item()
    assert zerr < Z_TOL, f"pre-VQ latents differ from the reference by {zerr:.2e}"
    flips = assert_ids_match_or_near_tie(ids, c.ids, z, c.sd["codebook.embeddings"], name)
    # tier (ii)
    recon = m.decode(c.ids.cuda(), c.is_image)
    err = (c.strided(recon.cpu()) - c.recon).abs().max().item()
    assert err < PIXEL_TOL, f"decode differs from the reference by {err:.2e} (|ref|max {c.recon_absmax:.2f})"
    psnr = orc.psnr(c.strided(recon.cpu()), c.recon)
    assert psnr > 80.0
    print(f"{name}: id flips {flips}, z err {zerr:.1e}, pixel err {err:.1e}, PSNR vs ref {psnr:.1f} dB")


# every arithmetic path of the engine: (gemm_mode, attn_mode, gemm_pl).  gemm_mode 1 (every GEMM on the bf16x3 kernel) is no
# longer a mode anybody selects: gemm_x3.hip stays in the product as the default mode's fallback for GEMMs without a known
# operand range ('l' / 'r' pooling Linears, the cnn patch-embed: fixtures var_pool_l / var_up_r / var_cnn run it inside the
# default mode) and keeps ONE fixture here as its whole-engine check.
ENGINE_MODES = [(2, 1, 1), (2, 1, 0), (2, 0, 1), (0, 0, 0)]
X3_ENGINE_CASE = "heavy_s2_sdpa_r64_vid"


@pytest.mark.parametrize("name,modes", [(n, m) for n in HEAVY_CASES for m in ENGINE_MODES] + [(X3_ENGINE_CASE, (1, 1, 0))],
                         ids=lambda v: v if isinstance(v, str) else "gemm%d_attn%d_pl%d" % v)
def test_heavy_statistics_vs_reference_golden(models, name, modes):
    """Weights with trained-checkpoint statistics (synth profile "heavy": Student-t weights, LayerNorm gains with
    outlier channels up to 20, q/k scales up to 4 = logits up to 128, large biases) on image-like and constant-colour
    inputs, against outputs of the reference itself -- the family that stresses the power-of-two operand scales of the
    fp16-split kernels (VERDICT r02 "missing" #2).  Outputs reach |pixel| = 53 here and the reference's OWN fp32 result
    is 1e-4 .. 1.5e-4 away from the fp64 one (stored in the fixture by tests/golden/make_golden.py), so the absolute
    bars of the standard fixtures (1e-4 / 2e-5) are below the reference's own rounding noise and the bars are stated
    in units of that noise: 3x for the fp32-MFMA and bf16x3 modes (measured 1.0-2.2x), 8x for the fp16-split default
    (measured 2.3-6.6x over the data-flow variants: its operands carry 22 significant bits, fp32's 24 -- an intrinsic
    factor ~3-4 that the standard fixtures, whose noise floor is 1e-6, never showed; profiles/r03_heavy_statistics_parity.txt).
    ids: no flip that is not a provable near-tie (observed: 0 in every mode)."""
    from omnitokenizer_amd import _lib, ops
    c = GoldenCase(name)
    m = models(c)
    gm, am, pl = modes
    try:
        _lib.set_option("gemm_mode", gm)
        _lib.set_option("attn_mode", am)
        _lib.set_option("gemm_pl", pl)
        x = c.x.cuda()
        ids, z = m.encode(x, c.is_image, return_latents=True)
        recon = m.decode(c.ids.cuda(), c.is_image)
    finally:
        _lib.set_option("gemm_mode", 2)
        _lib.set_option("attn_mode", 1)
        _lib.set_option("gemm_pl", 1)
    assert torch.equal(ops.vq_argmin(c.z.cuda(), m.codebook.embeddings.data).cpu(), c.ids)
    zerr = (z.cpu() - c.z).abs().max().item()
    flips = assert_ids_match_or_near_tie(ids, c.ids, z, c.sd["codebook.embeddings"], name)
    err = (c.strided(recon.cpu()) - c.recon).abs().max().item()
    print(f"{name} {modes}: id flips {flips}, z err {zerr:.1e} (reference fp32 noise {c.fp32_noise_z:.1e}), pixel err {err:.1e} "
          f"(noise {c.fp32_noise_pix:.1e}, |ref|max {c.recon_absmax:.1f})")
    # bars = the largest ratio measured over the five fixtures and the data-flow variants + 20 %: 6.6 -> 8.0 (fp16 split),
    # 2.2 -> 2.7 (fp32-MFMA and bf16x3 modes); profiles/r03_heavy_statistics_parity.txt, profiles/r04_heavy_batch_parity.txt
    k = 8.0 if gm == 2 else 2.7
    assert zerr < max(Z_TOL, k * c.fp32_noise_z), f"pre-VQ latents differ from the reference by {zerr:.2e}"
    assert err < max(PIXEL_TOL, k * c.fp32_noise_pix), f"decode differs from the reference by {err:.2e}"
    # absolute caps next to the noise-relative ones (z is unit-norm; pixels relative to the fixture's range)
    assert zerr < 1.5e-4 and err < 3e-5 * max(1.0, c.recon_absmax), (zerr, err, c.recon_absmax)
    assert torch.isfinite(recon).all()


def heavy_batch_report(c, ids, z, recon, noise_ratio_bar):
    """Batch-scale result in the reference's own currency (fixture heavy_*_b8: the reference run in fp32 and in fp64).
    The fixture stores, per token, the L2 distance of the fp64 latent to the nearest boundary of its code's cell: a
    perturbation dz can flip the id only if |dz|_2 >= that distance.  Raises on an id flip (against the fp64 ids)
      * that this run's own latent error at that token does not explain (the quantiser would be wrong), or
      * whose boundary distance exceeds `noise_ratio_bar` x the reference's own largest fp32-vs-fp64 latent distance
        (the flip would not be attributable to arithmetic noise of the accepted size).
    The slack term covers the rounding of the fp32 distance evaluation itself ((xx - 2 dot) + ee at magnitudes up to
    |e|^2 ~ 30, codebook.py:82-84: ~4e-6 on a distance gap, divided by 2 |e_a - e_b| ~ 8 for a randn codebook)."""
    ids, ids32, ids64 = ids.reshape(-1).cpu(), c.ids.reshape(-1), c.ids64.reshape(-1)
    z = z.reshape(-1, z.shape[-1]).cpu().double()
    z32, z64 = c.z.reshape(-1, 8).double(), c.z64.reshape(-1, 8)
    n = ids.numel()
    ours_l2 = (z - z64).norm(dim=1)          # per-token distance to the exact latent
    ref_l2 = (z32 - z64).norm(dim=1)         # ... of the reference's own fp32 run
    B = c.batch
    z_clip = (z - z32).abs().reshape(B, -1).amax(1)
    noise = c.fp32_noise_l2_max
    out = dict(tokens=n, flips_vs_ref32=int((ids != ids32).sum()), flips_vs_ref64=int((ids != ids64).sum()),
               ref_own_flips=int((ids32 != ids64).sum()),
               # tokens a run COULD flip: boundary distance below its own latent error there
               at_risk_ours=int((c.boundary < ours_l2).sum()), at_risk_ref=int((c.boundary < ref_l2).sum()),
               at_risk_1x_noise=int((c.boundary < noise).sum()),
               l2_noise_ratio_max=float(ours_l2.max() / ref_l2.max()),
               l2_noise_ratio_median=float(ours_l2.median() / ref_l2.median()),
               l2_noise_ratio_p999=float(ours_l2.quantile(0.999) / ref_l2.quantile(0.999)),
               z_err_over_noise_clip=[round(float(a / b), 2) for a, b in zip(z_clip, c.fp32_noise_z_clip)],
               z_err=float(z_clip.max()))
    slack = 2e-6
    bad = (ids != ids64).nonzero().flatten().tolist()
    out["flips"] = [(i, f"boundary {c.boundary[i]:.2e}", f"own error {ours_l2[i]:.2e}", f"reference's {ref_l2[i]:.2e}") for i in bad]
    if recon is not None:
        rec = c.strided(recon.cpu())
        pix_clip = (rec - c.recon).abs().reshape(B, -1).amax(1)
        out["pix_err"] = float(pix_clip.max())
        out["pix_err_over_noise_clip"] = [round(float(a / b), 2) for a, b in zip(pix_clip, c.fp32_noise_pix_clip)]
    for i in bad:
        assert c.boundary[i] <= ours_l2[i] + slack, f"token {i}: flipped although its latent error {ours_l2[i]:.2e} is inside the cell ({c.boundary[i]:.2e})"
        assert c.boundary[i] <= noise_ratio_bar * noise + slack, \
            f"token {i}: flip at boundary distance {c.boundary[i]:.2e} > {noise_ratio_bar} x reference noise {noise:.2e}"
    return out


# measured on the MI355X (profiles/r04_heavy_batch_parity.txt), bars = measured + 20 %: (gemm_mode, attn_mode, gemm_pl) ->
# (max over tokens of |z - z_fp64|_2 relative to the reference's own fp32 run, max |pixel - ref| / reference noise)
HEAVY_BATCH_BARS = {(2, 1, 1): (4.1, 1.0), (0, 0, 0): (2.3, 1.0)}


@pytest.mark.parametrize("modes", [(2, 1, 1), (0, 0, 0)], ids=lambda m: "gemm%d_attn%d_pl%d" % m)
def test_heavy_statistics_at_batch_scale(models, modes):
    """40 960 tokens (8 distinct 17x256x256 clips in ONE encode / decode, clip 0 a constant colour) on the heavy-tailed
    weight profile, every arithmetic mode, against the reference run in fp32 and in fp64: id flips are counted against
    both id sets next to the reference's own fp32-vs-fp64 flips, every flip must be explained by a latent perturbation
    of the mode's measured noise multiple, and the per-token latent noise is stated as a multiple of the reference's."""
    from omnitokenizer_amd import _lib
    c = GoldenCase(HEAVY_BATCH_CASE)
    m = models(c)
    gm, am, pl = modes
    try:
        _lib.set_option("gemm_mode", gm)
        _lib.set_option("attn_mode", am)
        _lib.set_option("gemm_pl", pl)
        ids, z = m.encode(c.x.cuda(), False, return_latents=True)
        recon = m.decode(c.ids.cuda(), False)
    finally:
        _lib.set_option("gemm_mode", 2)
        _lib.set_option("attn_mode", 1)
        _lib.set_option("gemm_pl", 1)
    zbar, pbar = HEAVY_BATCH_BARS[modes]
    r = heavy_batch_report(c, ids, z, recon, zbar)
    per1e5 = 1e5 / r["tokens"]
    print(f"{c.name} {modes}: flips vs the reference's fp64 ids {r['flips_vs_ref64']} ({r['flips_vs_ref64'] * per1e5:.1f} per 1e5 tokens; "
          f"tokens whose cell boundary is closer than this run's own latent error: {r['at_risk_ours']}), vs its fp32 ids "
          f"{r['flips_vs_ref32']}; the reference's own fp32-vs-fp64 flips {r['ref_own_flips']} (at risk: {r['at_risk_ref']}; within 1x "
          f"its largest noise: {r['at_risk_1x_noise']}); per-token |z - z64|_2 as a multiple of the reference's: median "
          f"{r['l2_noise_ratio_median']:.2f}, p99.9 {r['l2_noise_ratio_p999']:.2f}, max {r['l2_noise_ratio_max']:.2f}; z err "
          f"{r['z_err']:.1e} = {r['z_err_over_noise_clip']} x noise per clip; pixel err {r['pix_err']:.1e} = "
          f"{r['pix_err_over_noise_clip']} x noise per clip (|ref|max {c.recon_absmax:.1f}); flips: {r['flips']}")
    # a run can only flip tokens whose boundary lies inside its own error (asserted per flip above); the count is bounded by
    # the tokens within the mode's accepted noise multiple of a boundary
    assert r["flips_vs_ref64"] <= int((c.boundary < zbar * c.fp32_noise_l2_max).sum()), r
    # ... and, principled rather than measured: no more flips than the tokens the REFERENCE's own fp32 run could flip (its
    # latent error there exceeds the boundary distance: 16 of 40 960) plus a small constant (observed: 0 in every mode)
    assert r["flips_vs_ref64"] <= r["at_risk_ref"] + 2, r
    assert max(r["l2_noise_ratio_max"], r["l2_noise_ratio_median"]) <= zbar, r
    assert max(r["pix_err_over_noise_clip"]) <= pbar, r
    assert torch.isfinite(recon).all()


@pytest.mark.parametrize("name", ["s2_sdpa_r64_vid", "s2_sdpa_r256_img", "heavy_s2_sdpa_r128_vid_16k"])
def test_window_attention_paths_agree(models, name):
    """'w' blocks (reference attention.py:254-293) on packed operands and the fp16 matrix cores ("attn_window_mode" 1, the
    default) against the fp32-MFMA window kernel (mode 0) and the reference's golden outputs: both satisfy the same bars."""
    from omnitokenizer_amd import _lib
    c = GoldenCase(name)
    m = models(c)
    res = {}
    try:
        for mode in (0, 1):
            _lib.set_option("attn_window_mode", mode)
            ids, z = m.encode(c.x.cuda(), c.is_image, return_latents=True)
            res[mode] = (ids.cpu(), z.cpu())
    finally:
        _lib.set_option("attn_window_mode", 1)
    tol = max(Z_TOL, 8.0 * c.fp32_noise_z)
    for mode, (ids, z) in res.items():
        zerr = (z - c.z).abs().max().item()
        print(f"{name} attn_window_mode {mode}: z err {zerr:.2e}, flips {(ids != c.ids).sum().item()}")
        assert zerr < tol, (mode, zerr)
        assert_ids_match_or_near_tie(ids, c.ids, z, c.sd["codebook.embeddings"], f"{name} window mode {mode}")
    assert (res[0][1] - res[1][1]).abs().max().item() < 2 * tol


def test_one_data_flow_at_every_size_and_the_option_still_selects_the_other(models):
    """r06: "pl_min_tokens" defaults to 0 -- calls of every size run the plane data flow (thin tiles for small calls, same bits as
    the 256 x 256 tiles), so 2 clips (10240 tokens) and 3 clips (15360) both equal an explicit plane-flow run bit for bit.  The
    option still routes calls below a threshold to the fp32-activation flow (12288 = what rounds 4-5 shipped): checked against
    that flow selected explicitly."""
    from omnitokenizer_amd import _lib
    c = GoldenCase(HEAVY_BATCH_CASE)
    m = models(c)
    x = c.x.cuda()
    assert _lib.get_option("pl_min_tokens") == int(os.environ.get("OMNITOK_TEST_PL_MIN_TOKENS", "0"))

    def run(n, pl, min_tokens):
        try:
            _lib.set_option("gemm_pl", pl)
            _lib.set_option("pl_min_tokens", min_tokens)
            ids, z = m.encode(x[:n], c.is_image, return_latents=True)
            return ids.cpu(), z.cpu(), m.decode(ids, c.is_image).cpu()
        finally:
            _lib.set_option("gemm_pl", 1)
            _lib.set_option("pl_min_tokens", int(os.environ.get("OMNITOK_TEST_PL_MIN_TOKENS", "0")))  # tests/conftest.py

    for n in (2, 3):   # the default: planes at both sizes
        d, e = run(n, 1, 0), run(n, 1, 0)
        for a_, b_ in zip(d, e):
            assert torch.equal(a_, b_)
    one, three = run(1, 1, 0), run(3, 1, 0)   # ... and a clip gets the same bits alone and inside a batch
    for a_, b_ in zip(one, three):
        assert torch.equal(a_, b_[:1])
    for n, planes in ((2, False), (3, True)):
        tokens = c.ids[:n].numel()
        assert (tokens >= 12288) == planes
        auto = run(n, 1, 12288)
        forced = run(n, 1 if planes else 0, 0)
        other = run(n, 0 if planes else 1, 0)
        assert all(torch.equal(a, b) for a, b in zip(auto, forced)), (n, planes)
        assert not torch.equal(auto[1], other[1])     # the two flows round differently: the switch is observable
        assert (auto[1] - other[1]).abs().max().item() < 2 * max(Z_TOL, 8.0 * c.fp32_noise_z)


def test_tile_schedule_does_not_show_in_the_engine_outputs(models):
    """r06: at 8 clips (40 960 tokens) the plane GEMM's size rule runs full rounds of 256 x 256 tiles plus a tail of thin tiles for the
    N = 512 / 1024 launches ("pl_tail" 1) and 16-row stats_pack workgroups; forcing one big-tile launch ("pl_tail" 0), thin tiles
    everywhere ("pl_cfg" 5 / 6) or the 64-row stats_pack form ("sp_small_blocks" 1) changes no bit of ids, latents or pixels."""
    from omnitokenizer_amd import _lib
    c = GoldenCase(HEAVY_BATCH_CASE)
    m = models(c)
    x = c.x.cuda()

    def run(**opts):
        try:
            for k, v in opts.items():
                _lib.set_option(k, v)
            ids, z = m.encode(x, c.is_image, return_latents=True)
            return ids.clone(), z.clone(), m.decode(ids, c.is_image).clone()
        finally:
            _lib.set_option("pl_tail", 1)
            _lib.set_option("pl_cfg", 0)
            _lib.set_option("sp_small_blocks", 0)
    base = run()
    for opts in (dict(pl_tail=0), dict(pl_cfg=5), dict(pl_cfg=6), dict(pl_cfg=1), dict(sp_small_blocks=1)):
        got = run(**opts)
        for a_, b_ in zip(base, got):
            assert torch.equal(a_, b_), opts
    # ... and one clip alone gets the bits it has inside the batch (another grid for every launch)
    ids1, z1 = m.encode(x[:1].contiguous(), c.is_image, return_latents=True)
    assert torch.equal(ids1, base[0][:1]) and torch.equal(z1, base[1][:1])
    assert torch.equal(m.decode(ids1, c.is_image), base[2][:1])


def test_prevq_fusion_and_temporal_chunks_are_bit_identical(models):
    """Three r05 changes that must not change a bit: "prevq_fuse" (pre_vq inside the encoder's last LayerNorm pass),
    "vq_screen" (the fp16-screened nearest-code search) and "temporal_chunk" (the temporal q|k|v GEMM + attention run chunk by chunk through an Infinity-Cache-sized buffer;
    process option and per-engine option).  8 distinct 17x256^2 clips, ragged last chunk (3 + 3 + 2) included."""
    from omnitokenizer_amd import _lib
    c = GoldenCase(HEAVY_BATCH_CASE)
    m = models(c)
    x = c.x.cuda()

    def run(**opts):
        try:
            for k, v in opts.items():
                _lib.set_option(k, v)
            ids, z = m.encode(x, c.is_image, return_latents=True)
            return ids.cpu(), z.cpu(), m.decode(ids, c.is_image).cpu()
        finally:
            _lib.set_option("prevq_fuse", 1)
            _lib.set_option("temporal_chunk", 0)
            _lib.set_option("vq_screen", 1)
            _lib.set_option("temporal_fused", 1)

    # the chunk option lives on the two-kernel temporal flow ("temporal_fused" 0): every arm below runs that flow
    _lib.set_option("temporal_fused", 0)
    base = run(prevq_fuse=0, temporal_chunk=0, vq_screen=0, temporal_fused=0)
    for opts in (dict(prevq_fuse=1, vq_screen=0), dict(vq_screen=1, prevq_fuse=0), dict(temporal_chunk=4), dict(temporal_chunk=3), dict(temporal_chunk=1),
                 dict(temporal_chunk=8), dict(temporal_chunk=100)):
        got = run(temporal_fused=0, **opts)
        assert all(torch.equal(a, b) for a, b in zip(base, got)), opts
    m._sync_engine()
    lib = _lib.load()
    try:
        assert lib.omnitok_engine_set_option(m._engine, b"temporal_chunk", 2) == 0
        got = run(temporal_fused=0)
    finally:
        lib.omnitok_engine_set_option(m._engine, b"temporal_chunk", -1)
    assert all(torch.equal(a, b) for a, b in zip(base, got))


def test_two_engines_with_different_modes_in_one_process():
    """The arithmetic / data-flow modes are per-engine fields (omnitok_engine_set_option): two modules of one process run
    different modes side by side, each bit-identical to a run of that mode selected process-wide."""
    from omnitokenizer_amd import OmniTokenizer_VQGAN, _lib
    c = GoldenCase("s2_sdpa_r64_vid")

    def build():
        m = OmniTokenizer_VQGAN(c.args, attention_mode=c.mode)
        m.load_state_dict(c.sd, strict=True)
        return m.cuda().eval()
    a, b = build(), build()
    x = c.x.cuda()
    _lib.set_option("pl_min_tokens", 0)  # the fixture is a small call: keep the plane flow the process default here
    a.set_option("gemm_mode", 0)     # fp32-input MFMA GEMMs ...
    a.set_option("attn_mode", 0)
    b.set_option("gemm_pl", 0)       # ... next to the fp16-split GEMMs with fp32 activations between the kernels
    za = a.encode(x, False, return_latents=True)[1]
    zb = b.encode(x, False, return_latents=True)[1]
    zd = build().encode(x, False, return_latents=True)[1]   # a third engine on the process defaults, interleaved
    za2 = a.encode(x, False, return_latents=True)[1]
    assert torch.equal(za, za2) and not torch.equal(za, zb) and not torch.equal(zb, zd)
    ref = build()
    try:
        _lib.set_option("gemm_mode", 0)
        _lib.set_option("attn_mode", 0)
        assert torch.equal(ref.encode(x, False, return_latents=True)[1], za)
        _lib.set_option("gemm_mode", 2)
        _lib.set_option("attn_mode", 1)
        _lib.set_option("gemm_pl", 0)
        assert torch.equal(ref.encode(x, False, return_latents=True)[1], zb)
        # a per-engine value wins over the process default; -1 hands the engine back to it
        assert torch.equal(a.encode(x, False, return_latents=True)[1], za)
        a.set_option("gemm_mode", -1)
        a.set_option("attn_mode", -1)
        assert torch.equal(a.encode(x, False, return_latents=True)[1], zb)
    finally:
        _lib.set_option("gemm_mode", 2)
        _lib.set_option("attn_mode", 1)
        _lib.set_option("gemm_pl", 1)
        _lib.set_option("pl_min_tokens", int(os.environ.get("OMNITOK_TEST_PL_MIN_TOKENS", "0")))
    with pytest.raises(ValueError):
        a.set_option("h2_tile", 3)   # a tuning knob of the stand-alone kernels, not an engine mode
    for m in (za, zb, zd):
        assert (m.cpu() - c.z).abs().max().item() < Z_TOL


def test_flat_ids_and_embeddings(models):
    c = GoldenCase("s2_sdpa_r64_vid")
    m = models(c)
    emb, ids = m.encode(c.x.cuda(), False, include_embeddings=True)
    assert tuple(emb.shape) == (c.batch, 8) + tuple(c.ids.shape[1:])
    same = (ids.cpu() == c.ids)
    e = emb.permute(0, 2, 3, 4, 1).cpu()
    assert (e[same] - c.emb[same]).abs().max().item() < 1e-6  # (e - z) + z straight-through value
    a = m.decode(c.ids.cuda(), False)
    b = m.decode(c.ids.reshape(c.batch, -1).cuda(), False)  # flat ids: h = w = resolution // patch
    assert torch.equal(a, b)
    ci = GoldenCase("s2_sdpa_r64_img")
    mi = models(ci)
    assert torch.equal(mi.decode(ci.ids.cuda(), True), mi.decode(ci.ids.reshape(ci.batch, -1).cuda(), True))
    assert mi.decode(ci.ids.cuda(), True).shape == (ci.batch, 3, 64, 64)


def test_forward_log_image(models):
    c = GoldenCase("s2_sdpa_r64_img")
    m = models(c)
    frames, frames_recon, x, x_recon, vq = m(c.x.cuda(), log_image=True)
    assert torch.equal(vq["encodings"].cpu(), c.ids)
    assert (c.strided(x_recon.cpu()) - c.recon).abs().max().item() < PIXEL_TOL


def test_forward_codebook_statistics(models):
    """batch_usage / perplexity / avg_usage / EMA buffer of reference Codebook.forward
    (modules/codebook.py:54-72, 122-140), which vqgan_eval.py:152,195 reads."""
    c = GoldenCase("s2_sdpa_r64_vid")
    from omnitokenizer_amd import OmniTokenizer_VQGAN
    m = OmniTokenizer_VQGAN(c.args)
    m.load_state_dict(c.sd, strict=True)
    m = m.cuda().eval()
    n_codes = c.cfg.n_codes
    ema = None
    for call in range(2):
        x = c.x.cuda() if call == 0 else (c.x * 0.5).cuda()
        _, _, _, _, vq = m(x, log_image=True)
        ids = vq["encodings"].cpu().reshape(-1)
        usage = torch.bincount(ids, minlength=n_codes).float() / ids.numel()
        avg_probs = usage
        perp = torch.exp(-torch.sum(avg_probs * torch.log(avg_probs + 1e-10)))
        ema = usage if call == 0 else 0.99 * ema + 0.01 * usage
        avg_usage = (ema > 1 / n_codes).sum() / n_codes
        assert (vq["batch_usage"].cpu() - usage).abs().max().item() < 1e-7
        assert abs(vq["perplexity"].item() - perp.item()) < 1e-3 * perp.item()
        assert abs(vq["avg_usage"].item() - avg_usage.item()) < 1e-6
        assert (m.codebook.codebook_usage.data.cpu() - ema).abs().max().item() < 1e-7
    assert m.codebook.call_cnt == 2


def test_encode_mutates_codebook_usage_like_the_reference():
    """Every encode() of the reference runs Codebook.forward, which rewrites the `codebook_usage` buffer and bumps
    `call_cnt` even in eval mode (codebook.py:122-143): state_dict() after N encodes must equal the reference's
    (tests/golden/usage_state_s2_sdpa_r64.npz, generated from the reference itself); the module flag switches it off."""
    from omnitokenizer_amd import OmniTokenizer_VQGAN, synth
    c = GoldenCase("s2_sdpa_r64_img")
    g = np.load(os.path.join(GOLDEN, "usage_state_s2_sdpa_r64.npz"))
    assert synth.state_checksum(c.sd) == int(g["state_crc"])
    m = OmniTokenizer_VQGAN(c.args)
    m.load_state_dict(c.sd, strict=True)
    m = m.cuda().eval()
    xi, xv = synth.synth_image(2, 64, seed=1234).cuda(), synth.synth_video(2, 5, 64, seed=1234).cuda()
    for i, (x, is_image) in enumerate(((xi, True), (xv, False), (xi, True))):
        ids = m.encode(x, is_image)
        assert torch.equal(ids.cpu(), torch.from_numpy(g[f"ids{i}"].astype(np.int64)))
        got = m.state_dict()["codebook.codebook_usage"].cpu()
        assert (got - torch.from_numpy(g["usage"][i])).abs().max().item() < 1e-7
    assert m.codebook.call_cnt == int(g["call_cnt"]) == 3
    # forward(log_image=True) is ONE Codebook.forward: one update, not two
    m(xi, log_image=True)
    assert m.codebook.call_cnt == 4
    # opt-out: encode() without side effects
    m.update_codebook_usage_on_encode = False
    before = m.state_dict()["codebook.codebook_usage"].clone()
    m.encode(xv, False)
    assert m.codebook.call_cnt == 4 and torch.equal(m.state_dict()["codebook.codebook_usage"], before)


def test_ckpt_parity_tool_on_the_gpu(tmp_path, capsys):
    """tools/ckpt_parity.py end to end on a synthetic PL-format checkpoint (heavy-tailed weights, image-like inputs): both
    arithmetic modes pass against the checker (the oracle here; the reference where it is mounted)."""
    import json
    from tests import ckpt_parity
    from tests.test_ckpt_parity_tool import make_ckpt
    ck = str(tmp_path / "synthetic.ckpt")
    make_ckpt(ck, resolution=64)
    assert ckpt_parity.main(["--ckpt", ck, "--synthetic", "2", "--frames", "5", "--batch", "2", "--oracle"]) == 0
    lines = [json.loads(l) for l in capsys.readouterr().out.strip().splitlines()]
    summ = {l["mode"]: l for l in lines if l["event"] == "summary"}
    assert set(summ) == {"default", "strict_fp32"}
    for s_ in summ.values():
        assert s_["pass"] and s_["not_near_tie"] == 0 and s_["pixel_err"] < PIXEL_TOL and s_["psnr_vs_checker"] > 80.0


def test_error_behaviour(models):
    c = GoldenCase("s2_sdpa_r64_vid")
    m = models(c)
    with pytest.raises(AssertionError):  # reference omnitokenizer.py:931: (f - 1) % pt == 0
        m.encode(torch.zeros(1, 3, 4, 64, 64, device="cuda"), False)
    with pytest.raises((ValueError, AssertionError)):
        m.encode(torch.zeros(1, 3, 5, 60, 60, device="cuda"), False)
    with pytest.raises(RuntimeError):  # no CPU fallback
        m.encode(torch.zeros(1, 3, 5, 64, 64), False)
    with pytest.raises(IndexError):          # like the reference's F.embedding, by default
        bad = c.ids.clone()
        bad[0, 0, 0, 0] = 9000
        m.decode(bad.cuda(), False)
    m.decode(bad.cuda(), False, check_ids=False)   # opt-out: maps to code 0 without a host read-back
    assert m.encode(torch.zeros(0, 3, 5, 64, 64, device="cuda"), False).shape == (0, 2, 8, 8)


def test_in_place_weight_edits_are_seen(models):
    """The engine holds its own copies of the weights; the change detector reads the (storage pointer, in-place version)
    of every parameter and buffer (cached tensor list, no state_dict() walk per call): an in-place edit of ANY tensor
    after the first encode re-uploads the weights without mark_weights_changed()."""
    from omnitokenizer_amd import OmniTokenizer_VQGAN
    c = GoldenCase("s2_sdpa_r64_img")
    m = OmniTokenizer_VQGAN(c.args, attention_mode=c.mode)
    m.load_state_dict(c.sd)
    m = m.cuda().eval()
    x = c.x.cuda()
    ids0 = m.encode(x, True)
    with torch.no_grad():
        getattr(m.pre_vq_conv, "1").weight.mul_(-1.0)   # not one of the old sentinel tensors
    ids1 = m.encode(x, True)
    assert not torch.equal(ids0, ids1)
    m.load_state_dict(c.sd)                 # reloading bumps the version by itself
    assert torch.equal(m.encode(x, True), ids0)
    bad = {("module." + k): v for k, v in c.sd.items()}
    with pytest.raises(RuntimeError, match="lacks"):
        m.load_state_dict(bad)


def test_degenerate_ff_layernorm_falls_back():
    """a FeedForward LayerNorm with gamma == beta == 0 has operand bound 0; the plane data flow (whose producers
    scale by constants derived from that bound) must hand the Transformer to the in-loop-split GEMMs instead of failing."""
    from omnitokenizer_amd import OmniTokenizer_VQGAN
    c = GoldenCase("s2_sdpa_r64_vid")
    sd = {k: v.clone() for k, v in c.sd.items()}
    sd["encoder.enc_spatial_transformer.layers.1.3.0.weight"].zero_()
    sd["encoder.enc_spatial_transformer.layers.1.3.0.bias"].zero_()
    sd["decoder.dec_temporal_transformer.layers.2.3.0.weight"].zero_()
    sd["decoder.dec_temporal_transformer.layers.2.3.0.bias"].zero_()
    m = OmniTokenizer_VQGAN(c.args, attention_mode=c.mode)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    with torch.no_grad():
        taps = {}
        ids_ref = orc.encode(sd, c.x, False, c.cfg, taps=taps)
        rec_ref = orc.decode(sd, ids_ref, False, c.cfg)
    ids, z = m.encode(c.x.cuda(), False, return_latents=True)
    assert (z.cpu() - taps["z"]).abs().max().item() < Z_TOL
    assert_ids_match_or_near_tie(ids, ids_ref, z, sd["codebook.embeddings"], "degenerate FF LayerNorm")
    assert (m.decode(ids_ref.cuda(), False).cpu() - rec_ref).abs().max().item() < PIXEL_TOL


def test_data_edits_need_mark_weights_changed():
    """edits through `.data` (p.data.copy_(), the EMA store / restore idiom) move neither the storage pointer
    nor p._version, so the change detector cannot see them -- the documented contract is mark_weights_changed()."""
    from omnitokenizer_amd import OmniTokenizer_VQGAN
    c = GoldenCase("s2_sdpa_r64_img")
    m = OmniTokenizer_VQGAN(c.args, attention_mode=c.mode)
    m.load_state_dict(c.sd)
    m = m.cuda().eval()
    x = c.x.cuda()
    ids0 = m.encode(x, True)
    w = getattr(m.pre_vq_conv, "1").weight
    saved = w.data.clone()
    w.data.copy_(-saved)                      # invisible to (data_ptr, _version) ...
    m.mark_weights_changed()                  # ... so the caller says so
    ids1 = m.encode(x, True)
    assert not torch.equal(ids0, ids1)
    w.data.copy_(saved)
    m.mark_weights_changed()
    assert torch.equal(m.encode(x, True), ids0)


def test_decode_trusts_own_ids_by_identity_not_address():
    """decode() skips the id-range read-back only for the very tensor object encode() returned.  A foreign
    tensor that lands on the same address with the same shape and version 0 (the caching allocator recycles the block)
    is checked like any other input: out-of-range ids raise like the reference's F.embedding."""
    from omnitokenizer_amd import OmniTokenizer_VQGAN
    c = GoldenCase("s2_sdpa_r64_img")
    m = OmniTokenizer_VQGAN(c.args, attention_mode=c.mode)
    m.load_state_dict(c.sd)
    m = m.cuda().eval()
    ids = m.encode(c.x.cuda(), True)
    shape, addr = tuple(ids.shape), ids.data_ptr()
    m.decode(ids, True)                       # own ids: trusted
    del ids
    foreign = torch.full(shape, 8192 + 5, dtype=torch.int64, device="cuda")   # very likely the recycled block
    print("recycled the freed block:", foreign.data_ptr() == addr)
    with pytest.raises(IndexError):
        m.decode(foreign, True)
    ids = m.encode(c.x.cuda(), True)
    ids[0, 0, 0, 0] = 9000                    # modified after encode(): no longer trusted
    with pytest.raises(IndexError):
        m.decode(ids, True)


def test_oracle_parity_other_shapes(models):
    """Shapes without a golden fixture, straight against the CPU oracle (9 frames at 128 px,
    stage-2)."""
    from omnitokenizer_amd import synth
    c = GoldenCase("s2_sdpa_r128_vid_16k")
    m = models(c)
    x = synth.synth_video(1, 9, 128, seed=99)
    with torch.no_grad():
        taps = {}
        ids_ref = orc.encode(c.sd, x, False, c.cfg, taps=taps)
        rec_ref = orc.decode(c.sd, ids_ref, False, c.cfg)
    ids, z = m.encode(x.cuda(), False, return_latents=True)
    assert (z.cpu() - taps["z"]).abs().max().item() < Z_TOL
    assert_ids_match_or_near_tie(ids, ids_ref, z, c.sd["codebook.embeddings"], "r128_9f")
    assert (m.decode(ids_ref.cuda(), False).cpu() - rec_ref).abs().max().item() < PIXEL_TOL


@pytest.mark.parametrize("overrides,mode", [
    (dict(embedding_dim=256, heads=4, spatial_depth=2, enc_block="tw", dec_block="wt", temporal_depth=1, ff_mult=2.0,
          n_codes=1024, causal_in_temporal_transformer=False, causal_in_peg=False, temporal_patch_size=2), "sdpa"),
    (dict(embedding_dim=768, heads=12, spatial_depth=1, enc_block="t", dec_block="t", temporal_depth=2, ff_mult=3.0,
          n_codes=2048, spatial_pos="rel"), "legacy"),
    (dict(embedding_dim=384, heads=6, spatial_depth=3, enc_block="wtw", dec_block="ttw", temporal_depth=1,
          n_codes=4096, l2_code=False, causal_in_peg=False), "sdpa"),
])
def test_other_architectures_vs_oracle(overrides, mode):
    """Hyper-parameters no released checkpoint uses (width, heads, depth, ff_mult, block strings,
    non-causal temporal attention / PEG, no l2 normalisation before the quantiser), straight against
    the CPU oracle -- which tests/test_oracle_vs_reference.py pins to the live reference."""
    from omnitokenizer_amd import OmniTokenizer_VQGAN, make_args, synth
    from omnitokenizer_amd.config import OmniTokConfig
    args = make_args(2, resolution=64, **overrides)
    cfg = OmniTokConfig.from_args(args, attention_mode=mode)
    sd = synth.synth_state_dict(cfg, seed=13)
    m = OmniTokenizer_VQGAN(args, attention_mode=mode)
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    m = m.cuda().eval()
    for is_image, x in ((True, synth.synth_image(2, 64, seed=5)), (False, synth.synth_video(2, 5, 64, seed=6))):
        with torch.no_grad():
            taps = {}
            ids_ref = orc.encode(sd, x, is_image, cfg, taps=taps)
            rec_ref = orc.decode(sd, ids_ref, is_image, cfg)
        ids, z = m.encode(x.cuda(), is_image, return_latents=True)
        scale = max(1.0, taps["z"].abs().max().item())
        assert (z.cpu() - taps["z"]).abs().max().item() < Z_TOL * scale
        assert_ids_match_or_near_tie(ids, ids_ref, z, sd["codebook.embeddings"], str(overrides))
        assert (m.decode(ids_ref.cuda(), is_image).cpu() - rec_ref).abs().max().item() < PIXEL_TOL


# ---- size-independent properties at BASELINE.json sizes (C2: B=64 images, C3: B=32 clips) -------
@pytest.mark.parametrize("is_image,batch", [(True, 64), (False, 32)])
def test_full_size_properties(models, is_image, batch):
    from omnitokenizer_amd import synth
    c = GoldenCase("s2_sdpa_r256_img" if is_image else "s2_sdpa_r256_vid")
    m = models(c)
    base = synth.synth_image(4, 256, seed=7) if is_image else synth.synth_video(4, 17, 256, seed=7)
    reps = batch // 4
    x = torch.cat([base] * reps).cuda()            # batch = 4 distinct items repeated
    ids = m.encode(x, is_image)
    T = 1 if is_image else 5
    assert tuple(ids.shape) == (batch, T, 32, 32) and int(ids.min()) >= 0 and int(ids.max()) < 8192
    # batch independence: every repeat of an item gets identical ids, equal to encoding it alone
    first = ids[:4]
    for r in range(1, reps):
        assert torch.equal(ids[4 * r:4 * r + 4], first)
    assert torch.equal(m.encode(x[:1].contiguous(), is_image), ids[:1])
    # decode: batch independence + flat ids + determinism
    rec = m.decode(ids, is_image)
    assert rec.shape == ((batch, 3, 256, 256) if is_image else (batch, 3, 17, 256, 256))
    assert torch.isfinite(rec).all()
    assert torch.equal(rec[4:8], rec[:4])
    # one data flow at every call size (r06): a 2-item call decodes to the same bits as those items inside the full batch
    # (OMNITOK_TEST_PL_MIN_TOKENS=12288, the A/B arm, puts the small call on the other flow: pixels then agree to rounding)
    min_tokens = int(os.environ.get("OMNITOK_TEST_PL_MIN_TOKENS", "0"))
    rec2 = m.decode(ids[:2].contiguous(), is_image)
    if (ids[:2].numel() >= min_tokens) == (ids.numel() >= min_tokens):
        assert torch.equal(rec2, rec[:2])
    else:
        assert (rec2 - rec[:2]).abs().max().item() < PIXEL_TOL
    assert torch.equal(m.decode(ids, is_image), rec)
    # the golden item (image case: same shape) is reproduced inside the big batch
    if is_image:
        xg = c.x.cuda()
        big = torch.cat([xg, x[: batch - xg.shape[0]]])
        ids_big, z_big = m.encode(big, is_image, return_latents=True)
        n = xg.shape[0]
        # same rule as everywhere: a flip against the reference's ids must be a provable near-tie in fp64
        assert_ids_match_or_near_tie(ids_big[:n], c.ids, z_big[:n], c.sd["codebook.embeddings"], "golden item inside the big batch")
    # encode -> decode -> encode round trip stays in range and is deterministic
    ids2 = m.encode(rec.contiguous(), is_image)
    assert torch.equal(ids2, m.encode(rec.contiguous(), is_image))


def test_c3_batch_of_32_distinct_clips_vs_reference(models):
    """BASELINE config C3 exactly as bench.py runs it: 32 DISTINCT 17x256x256 clips in one encode / decode;
    the first, two interior and the last clip are compared with the reference's own outputs on those clips
    (tests/golden/s2_sdpa_r256_vid17_b32.npz, generated by make_golden.run_b32_case)."""
    import zlib
    import numpy as np
    from omnitokenizer_amd import synth
    g = np.load(os.path.join(GOLDEN, "s2_sdpa_r256_vid17_b32.npz"))
    c = GoldenCase("s2_sdpa_r256_vid17")     # same architecture / weights (seed 0)
    assert int(g["state_crc"]) == synth.state_checksum(c.sd)
    m = models(c)
    x = synth.synth_video(32, 17, 256, seed=int(g["input_seed"]))
    assert zlib.crc32(x.numpy().tobytes()) == int(g["input_crc"]), "synthetic input drifted"
    clips = [int(v) for v in g["clips"]]
    ids_ref = torch.from_numpy(g["ids"].astype(np.int64))
    z_ref, rec_ref, stride = torch.from_numpy(g["z"]), torch.from_numpy(g["recon"]), int(g["stride"])
    ids, z = m.encode(x.cuda(), False, return_latents=True)
    assert tuple(ids.shape) == (32, 5, 32, 32)
    zerr = (z[clips].cpu() - z_ref).abs().max().item()
    assert zerr < Z_TOL, f"pre-VQ latents of clips {clips} differ from the reference by {zerr:.2e}"
    flips = assert_ids_match_or_near_tie(ids[clips], ids_ref, z[clips], c.sd["codebook.embeddings"], "c3_b32")
    # decode the whole batch from ids that carry the reference's ids in the checked slots
    ids_in = ids.clone()
    ids_in[clips] = ids_ref.cuda()
    rec = m.decode(ids_in, False)
    err = (rec[clips][..., ::stride, ::stride].cpu() - rec_ref).abs().max().item()
    assert err < PIXEL_TOL, f"decoded pixels of clips {clips} differ from the reference by {err:.2e}"
    # all 32 clips are distinct and each one's result is independent of the batch
    assert len({zlib.crc32(ids[b].cpu().numpy().tobytes()) for b in range(32)}) == 32
    assert torch.equal(m.encode(x[20:21].cuda().contiguous(), False), ids[20:21])
    print(f"C3 batch of 32 distinct clips: id flips {flips}, z err {zerr:.1e}, pixel err {err:.1e}")


def test_c3_batch_all_32_clips_vs_oracle(models):
    """Every one of the 32 distinct clips bench.py feeds at C3 (not only the four the reference fixture holds), encoded and
    decoded in ONE batch on the MI355X, against the CPU oracle run clip by clip on the host (the oracle is pinned to the
    reference on four of these very clips by tests/test_oracle_vs_golden.py / the fixture above)."""
    import os
    from omnitokenizer_amd import synth
    c = GoldenCase("s2_sdpa_r256_vid17")
    m = models(c)
    x = synth.synth_video(32, 17, 256, seed=1234)
    ids, z = m.encode(x.cuda(), False, return_latents=True)
    ids_c, z_c = ids.cpu(), z.cpu()
    threads = torch.get_num_threads()
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    zerr = perr = 0.0
    flips = 0
    ids_ref_all = []
    try:
        with torch.no_grad():
            for b in range(32):
                taps = {}
                ids_ref = orc.encode(c.sd, x[b:b + 1], False, c.cfg, taps=taps)
                ids_ref_all.append(ids_ref)
                zerr = max(zerr, (z_c[b:b + 1] - taps["z"]).abs().max().item())
                flips += assert_ids_match_or_near_tie(ids_c[b:b + 1], ids_ref, z_c[b:b + 1], c.sd["codebook.embeddings"], f"clip {b}")
            rec = m.decode(torch.cat(ids_ref_all).cuda(), False).cpu()
            for b in range(32):
                rec_ref = orc.decode(c.sd, ids_ref_all[b], False, c.cfg)
                perr = max(perr, (rec[b:b + 1] - rec_ref).abs().max().item())
    finally:
        torch.set_num_threads(threads)
    print(f"C3, all 32 clips vs the oracle: id flips {flips}/{ids.numel()}, z err {zerr:.1e}, pixel err {perr:.1e}")
    assert zerr < Z_TOL and perr < PIXEL_TOL


def test_c5_long_sequence_stress(models):
    """BASELINE config C5 shape: one 65-frame 512x512 clip, n_codes = 16384 (T' = 17 -> streaming
    temporal kernel, N = 4096 tokens per frame -> 64 K/V tiles per query block, 64x64 window grid),
    against the CPU oracle."""
    from omnitokenizer_amd import OmniTokenizer_VQGAN, make_args, synth
    from omnitokenizer_amd.config import OmniTokConfig
    args = make_args(2, resolution=512, n_codes=16384, sequence_length=65)
    cfg = OmniTokConfig.from_args(args)
    sd = synth.synth_state_dict(cfg, seed=0)
    m = OmniTokenizer_VQGAN(args)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    x = synth.synth_video(1, 65, 512, seed=3)
    ids, z = m.encode(x.cuda(), False, return_latents=True)
    assert tuple(ids.shape) == (1, 17, 64, 64)
    with torch.no_grad():
        taps = {}
        ids_ref = orc.encode(sd, x, False, cfg, taps=taps)
        rec_ref = orc.decode(sd, ids_ref, False, cfg)
    zerr = (z.cpu() - taps["z"]).abs().max().item()
    assert zerr < 3e-5, zerr
    flips = assert_ids_match_or_near_tie(ids, ids_ref, z, sd["codebook.embeddings"], "c5")
    rec = m.decode(ids_ref.cuda(), False)
    assert rec.shape == (1, 3, 65, 512, 512)
    err = (rec.cpu() - rec_ref).abs().max().item()
    assert err < PIXEL_TOL, err
    # flat video ids need args.resolution = 512 (reference omnitokenizer.py:283-286)
    assert torch.equal(m.decode(ids_ref.reshape(1, -1).cuda(), False), rec)
    print(f"c5: id flips {flips}/{ids.numel()}, z err {zerr:.1e}, pixel err {err:.1e}")


# ---- --use_vae (reference omnitokenizer.py:260-266, 293-317; what DiT/Latte call) --------------
VAE_Z_TOL = 5e-5  # |z| reaches ~8 (mean + std*noise with std up to e^1.8); relative 1e-5


@pytest.mark.parametrize("name", VAE_CASES)
def test_vae_encode_decode_vs_reference_golden(models, name):
    c = GoldenCase(name)
    m = models(c)
    assert m.use_vae
    x = c.x.cuda()
    # (a) with the reference's stored noise
    z, mom = m.encode(x, c.is_image, noise=c.noise, return_moments=True)
    z5 = z.unsqueeze(2) if c.is_image else z
    assert tuple(z5.shape) == tuple(c.z.shape)
    merr = (mom.cpu() - c.moments).abs().max().item()
    zerr = (z5.cpu() - c.z).abs().max().item()
    assert merr < 2e-5, f"posterior moments differ from the reference by {merr:.2e}"
    assert zerr < VAE_Z_TOL, f"posterior sample differs from the reference by {zerr:.2e}"
    # (b) seeded like a reference user would: torch.manual_seed + host draw (modules/vae.py:16)
    torch.manual_seed(c.noise_seed)
    z_seeded = m.encode(x, c.is_image)
    assert torch.equal(z_seeded, z)
    # (c) posterior mode == mean
    mode = m.encode(x, c.is_image, sample_posterior=False)
    mode5 = mode.unsqueeze(2) if c.is_image else mode
    assert (mode5.cpu() - c.moments[:, :8]).abs().max().item() < 2e-5
    # (d) decode of the reference's z in the layouts the reference accepts
    recon = m.decode(c.decode_input().cuda(), c.is_image)
    err = (c.strided(recon.cpu()) - c.recon).abs().max().item()
    assert err < PIXEL_TOL, f"decode differs from the reference by {err:.2e}"
    flat = c.z.permute(0, 2, 3, 4, 1).reshape(c.z.shape[0], -1, 8).cuda()
    if c.is_image or c.cfg.resolution // c.cfg.patch_size == c.z.shape[-1]:
        assert torch.equal(m.decode(flat, c.is_image), recon)
    print(f"{name}: moments err {merr:.1e}, z err {zerr:.1e}, pixel err {err:.1e}")


def test_vae_forward_and_mode_errors(models):
    c = GoldenCase("vae_s2_sdpa_r64_vid")
    m = models(c)
    x = c.x.cuda()
    torch.manual_seed(7)
    frames, frames_recon, xx, x_recon, vq = m(x, log_image=True)
    assert vq is None and x_recon.shape == x.shape and frames.shape == frames_recon.shape
    torch.manual_seed(7)
    z = m.encode(x, False)
    assert torch.equal(m.decode(z.permute(0, 2, 3, 4, 1), False), x_recon)
    # encode()'s channel-first video output is NOT what decode() takes (reference quirk, :313-314)
    with pytest.raises((ValueError, RuntimeError)):
        m.decode(z, False)
    with pytest.raises(ValueError):
        m.encode(x, False, noise=torch.zeros(1, 8, 1, 8, 8))
    # the native VQ entry points refuse a use_vae engine and vice versa
    import ctypes
    from omnitokenizer_amd import _lib
    lib = _lib.load()
    ids = torch.zeros(2, 2, 8, 8, dtype=torch.int64, device="cuda")
    out = torch.empty(2, 3, 5, 64, 64, device="cuda")
    rc = lib.omnitok_decode(m._engine, ctypes.c_void_p(ids.data_ptr()), 2, 2, 8, 8, ctypes.c_void_p(out.data_ptr()), None)
    assert rc != 0 and b"use_vae" in lib.omnitok_last_error()
    cq = GoldenCase("s2_sdpa_r64_vid")
    mq = models(cq)
    mq.encode(cq.x.cuda(), False)
    rc = lib.omnitok_decode_vae(mq._engine, ctypes.c_void_p(out.data_ptr()), 0, 2, 2, 8, 8,
                                ctypes.c_void_p(out.data_ptr()), None)
    assert rc != 0 and b"use_vae" in lib.omnitok_last_error()


def test_vae_full_size_round_trip(models):
    """C3-sized clip batch slice through the VAE path: determinism, batch invariance and
    reconstruction == decode(encode) with the oracle on one clip."""
    from omnitokenizer_amd import synth
    c = GoldenCase("vae_s2_sdpa_r256_vid")
    m = models(c)
    x = synth.synth_video(4, 17, 256, seed=99).cuda()
    noise = torch.randn(4, 8, 5, 32, 32, generator=torch.Generator().manual_seed(5))
    z = m.encode(x, False, noise=noise)
    assert torch.isfinite(z).all()
    z_again = m.encode(x, False, noise=noise)
    assert torch.equal(z, z_again)
    z1 = m.encode(x[1:2], False, noise=noise[1:2])
    assert (z1 - z[1:2]).abs().max().item() < 1e-4  # batch slices see different tile schedules only
    rec = m.decode(z.permute(0, 2, 3, 4, 1), False)
    assert tuple(rec.shape) == (4, 3, 17, 256, 256) and torch.isfinite(rec).all()
    with torch.no_grad():
        z_ref = orc.encode_vae(c.sd, x[:1].cpu(), False, c.cfg, noise=noise[:1])
        rec_ref = orc.decode_vae(c.sd, z[:1].permute(0, 2, 3, 4, 1).cpu(), False, c.cfg)
    assert (z[:1].cpu() - z_ref).abs().max().item() < VAE_Z_TOL
    assert (rec[:1].cpu() - rec_ref).abs().max().item() < PIXEL_TOL


def test_hip_graph_capture_and_side_stream(models):
    """The engine launches everything on the caller's stream with no hidden synchronisation once
    its caches are warm, so encode+decode can be captured into one HIP graph (the launch-bound
    small-batch case) and can run on a non-default stream."""
    c = GoldenCase("s2_sdpa_r64_vid")
    m = models(c)
    x = c.x.cuda()
    ids0 = m.encode(x, False)
    rec0 = m.decode(ids0, False)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ids_s = m.encode(x, False)
        rec_s = m.decode(ids_s, False)
    torch.cuda.current_stream().wait_stream(s)
    assert torch.equal(ids_s, ids0) and torch.equal(rec_s, rec0)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ids_g = m.encode(x, False)
        rec_g = m.decode(ids_g, False)
    x2 = torch.roll(x, 1, dims=0).contiguous()
    ids2 = m.encode(x2, False)
    rec2 = m.decode(ids2, False)
    x.copy_(x2)  # new input in the captured buffer
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(ids_g, ids2) and torch.equal(rec_g, rec2)
    assert torch.equal(ids2.cpu(), torch.roll(c.ids, 1, dims=0))
    # a LARGER eager call afterwards makes the workspace grow; the captured graph still holds the old block's addresses,
    # so that block is kept alive (not handed back to the caching allocator) and a replay stays correct
    big = torch.cat([x2] * 4).contiguous()
    ids_big = m.encode(big, False)
    m.decode(ids_big, False)
    scratch = [torch.full((1 << 22,), 7.0, device="cuda") for _ in range(8)]  # would land in a recycled block
    x.copy_(x2)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(ids_g, ids2) and torch.equal(rec_g, rec2)
    del scratch


# ---- --use_external_codebook: VectorQuantize / cosine similarity -----------------
@pytest.mark.parametrize("name", EXT_CASES)
def test_external_codebook_vs_reference_golden(models, name):
    c = GoldenCase(name)
    m = models(c)
    assert m.use_external_codebook
    from omnitokenizer_amd import ops
    E = c.sd["codebook._codebook.embed"][0]
    x = c.x.cuda()
    emb, ids, z = m.encode(x, c.is_image, include_embeddings=True, return_latents=True)
    # the quantiser alone on the reference's z: bit-exact first-argmax (cosine) / first-argmin of cdist
    quant = ops.vq_argmax_cos if c.cfg.l2_code else ops.vq_argmin_cdist
    assert torch.equal(quant(c.z.cuda(), E.cuda()).cpu(), c.ids)
    zerr = (z.cpu() - c.z).abs().max().item()
    assert zerr < Z_TOL
    flips = assert_ids_match_or_near_tie(ids, c.ids, z, E, name)  # cosine, unit norm: nearest == most similar
    assert tuple(emb.shape) == (c.batch, 512) + tuple(c.ids.shape[1:])
    if flips == 0:
        assert (emb.permute(0, 2, 3, 4, 1)[..., ::8].cpu() - c.emb).abs().max().item() < 1e-5
    recon = m.decode(c.ids.cuda(), c.is_image)  # = the reference forward()'s decoder(project_out(embed[ids]))
    err = (c.strided(recon.cpu()) - c.recon).abs().max().item()
    assert err < PIXEL_TOL
    out = m(x, log_image=True)[4]
    assert set(out) == {"embeddings", "encodings", "commitment_loss", "perplexity", "avg_usage", "batch_usage"}
    if flips == 0:
        assert abs(float(out["perplexity"]) - c.perplexity) < 1e-3 * c.perplexity
    print(f"{name}: id flips {flips}, z err {zerr:.1e}, pixel err {err:.1e}")


def test_workspace_is_torch_memory_and_the_c_abi_still_owns_one():
    """The Python mirror lends the engine a block of PyTorch's caching allocator (sized by
    omnitok_engine_workspace_need_*); the bare C ABI keeps its own grow-only buffers.  Same results either way,
    and encode / decode run through the registered operators."""
    import ctypes
    from omnitokenizer_amd import OmniTokenizer_VQGAN, _lib
    c = GoldenCase("s2_sdpa_r64_vid")
    m = OmniTokenizer_VQGAN(c.args, attention_mode=c.mode)
    m.load_state_dict(c.sd, strict=True)
    m = m.cuda().eval()
    x = c.x.cuda()
    before = torch.cuda.memory_allocated()
    ids = m.encode(x, False)
    assert torch.equal(ids.cpu(), c.ids)
    lib = _lib.load()
    need = lib.omnitok_engine_workspace_need_encode(m._engine, *[x.shape[i] for i in (0, 2, 3, 4)])
    assert need > 0 and m._workspace is not None and m._workspace.numel() >= need
    assert torch.cuda.memory_allocated() - before >= need             # visible to torch's accounting
    assert lib.omnitok_engine_workspace_bytes(m._engine) <= m._workspace.numel()
    ids_op, _, _ = torch.ops.omnitok.vqgan_encode(x, m._handle, False, False)
    assert torch.equal(ids_op, ids)
    rec = m.decode(ids, False)
    assert torch.equal(torch.ops.omnitok.vqgan_decode(ids, m._handle), rec)
    # a too-small caller block is an error, not an overrun
    small = torch.empty(4096, dtype=torch.uint8, device="cuda")
    assert lib.omnitok_engine_set_workspace(m._engine, ctypes.c_void_p(small.data_ptr()), small.numel()) == 0
    out = torch.empty_like(ids)
    rc = lib.omnitok_encode(m._engine, ctypes.c_void_p(x.data_ptr()), *[x.shape[i] for i in (0, 2, 3, 4)],
                            ctypes.c_void_p(out.data_ptr()), None, None, torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and b"too small" in lib.omnitok_last_error()
    # back to the engine's own allocation (what a C caller gets by default)
    assert lib.omnitok_engine_set_workspace(m._engine, None, 0) == 0
    m._workspace = None
    rc = lib.omnitok_encode(m._engine, ctypes.c_void_p(x.data_ptr()), *[x.shape[i] for i in (0, 2, 3, 4)],
                            ctypes.c_void_p(out.data_ptr()), None, None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert rc == 0 and torch.equal(out, ids)
    assert lib.omnitok_engine_workspace_bytes(m._engine) > 0
