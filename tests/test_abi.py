"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/omnitok.h
declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from omnitokenizer_amd import build, _lib
    build.build()
    return _lib.load()


def test_header_symbols_exported(lib):
    from omnitokenizer_amd import _lib
    hdr = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("omnitok.h", "omnitok_lm.h", "omnitok_debug.h", "omnitok_comm.h"))
    # the measurement-only entry points live in their own header, outside the drop-in boundary
    assert "omnitok_debug_" not in open(os.path.join(ROOT, "include", "omnitok.h")).read()
    declared = set(re.findall(r"\b(omnitok_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"omnitok_stream_t"}
    assert len(declared) >= 30
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"{name} declared in omnitok.h but not exported"
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)


def test_version_and_argument_errors_without_gpu(lib):
    assert b"gfx950" in lib.omnitok_version()
    # argument validation happens before any HIP call, so it is testable on the CPU
    rc = lib.omnitok_gemm(None, 0, None, 0, None, None, 0, None, 0, 1, 1, 1, 0, 0, 0, 0, None)
    assert rc == -1 and b"null" in lib.omnitok_last_error()
    rc = lib.omnitok_rope_table(0, 64, 10000.0, None, None)
    assert rc == -1


def test_rope_table_host_matches_oracle(lib):
    import torch
    from omnitokenizer_amd import ops
    from oracle import omnitok_oracle as orc
    for n in (64, 1024):
        cos, sin = ops.rope_table(n)
        rc, rs = orc.rope_table(n)
        assert (cos - rc).abs().max().item() < 1e-6 and (sin - rs).abs().max().item() < 1e-6


def test_product_refuses_cpu_tensors(lib):
    import torch
    from omnitokenizer_amd import OmniTokenizer_VQGAN, make_args, ops
    with pytest.raises(RuntimeError, match="no CPU"):
        ops.layernorm(torch.zeros(4, 512), torch.ones(512))
    m = OmniTokenizer_VQGAN(make_args(2, resolution=64)).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.encode(torch.zeros(1, 3, 64, 64), True)


def test_state_dict_contract(lib):
    """Same key set / shapes as the reference's path; off-path keys are ignored on load."""
    import torch
    from omnitokenizer_amd import OmniTokenizer_VQGAN, make_args, synth
    from omnitokenizer_amd.config import OmniTokConfig
    for stage in (1, 2):
        args = make_args(stage, resolution=64)
        cfg = OmniTokConfig.from_args(args)
        m = OmniTokenizer_VQGAN(args)
        spec = synth.path_state_spec(cfg)
        sd = m.state_dict()
        assert list(sd.keys()) == list(spec.keys())
        assert all(tuple(sd[k].shape) == tuple(spec[k]) for k in spec)
        src = synth.synth_state_dict(cfg, seed=1)
        src["video_discriminator.main.0.weight"] = torch.zeros(3)
        src["perceptual_model.net.x"] = torch.zeros(1)
        res = m.load_state_dict(src, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        assert torch.equal(m.codebook.embeddings.data, src["codebook.embeddings"])
        assert m.codebook.n_codes == cfg.n_codes and m.n_codes == cfg.n_codes and not m.use_vae
        assert m.encoder.image_size == (64, 64)


def test_config_rejects_unbuilt_variants():
    from omnitokenizer_amd import make_args
    from omnitokenizer_amd.config import OmniTokConfig
    # in the DECODER 'n'/'r' Up blocks make the reference itself raise (omnitokenizer.py:1078; in the encoder
    # they run and are built, see below); GroupNorm(32) over the cnn decoder's 3 channels cannot be
    # constructed (base.py:274); the external VectorQuantize exists for codebook_type 'vq' only
    # (omnitokenizer.py:139-140)
    for bad in (dict(dec_block="tntt"), dict(dec_block="trtt"), dict(dec_block="tatt"), dict(enc_block="ttxw"),
                dict(patch_embed="cnn", norm_type="group"), dict(patch_embed="conv"),
                dict(use_external_codebook=True, codebook_type="lfq"),
                dict(use_external_codebook=True, use_vae=True), dict(dim_head=32)):
        with pytest.raises((NotImplementedError, ValueError)):
            OmniTokConfig.from_args(make_args(2, **bad))
    assert OmniTokConfig.from_args(make_args(2, use_vae=True)).use_vae
    c = OmniTokConfig.from_args(make_args(2, defer_spatial_pool=True, defer_temporal_pool=True, gen_upscale=2))
    assert (c.enc_patch_size, c.enc_temporal_patch_size, c.dec_patch_size, c.enc_grid_divisor) == (4, 2, 8, 2)
    c = OmniTokConfig.from_args(make_args(2, patch_embed="cnn", defer_spatial_pool=True, enc_block="tawl"))
    assert (c.enc_patch_size, c.dec_patch_size, c.enc_grid_divisor) == (8, 8, 4)  # defer_* ignored for cnn
    # encoder-side Up blocks (reference attention.py:116-150, 640-645): enc_block="ttnw" runs in the reference
    c = OmniTokConfig.from_args(make_args(2, enc_block="trnw"))
    assert (c.enc_grid_multiplier, c.enc_grid_divisor) == (4, 1)


def test_vae_state_dict_contract(lib):
    """--use_vae: pre_vq_conv emits mean|logvar (reference omnitokenizer.py:149-153)."""
    import torch
    from omnitokenizer_amd import OmniTokenizer_VQGAN, make_args
    m = OmniTokenizer_VQGAN(make_args(2, resolution=64, use_vae=True))
    assert m.use_vae
    assert tuple(m.state_dict()["pre_vq_conv.1.weight"].shape) == (16, 512)
    assert tuple(m.state_dict()["pre_vq_conv.1.bias"].shape) == (16,)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.encode(torch.zeros(1, 3, 64, 64), True)


def test_load_from_pl_checkpoint(tmp_path, lib):
    """PL-style checkpoints ({state_dict, hyper_parameters: {args}}, reference omnitokenizer.py:208,
    download.py:49) load through the drop-in class, off-path entries dropped, stale Namespaces
    back-filled like the reference's hasattr() defaults (omnitokenizer.py:70-98)."""
    import argparse
    import torch
    from omnitokenizer_amd import OmniTokenizer_VQGAN, make_args, synth
    from omnitokenizer_amd.config import OmniTokConfig
    args = make_args(1, resolution=64)
    cfg = OmniTokConfig.from_args(args)
    sd = dict(synth.synth_state_dict(cfg, seed=7))
    sd["image_discriminator.main.0.weight"] = torch.zeros(3)
    sd["perceptual_model.net.slice1.0.weight"] = torch.zeros(3)
    old = argparse.Namespace(**{k: v for k, v in vars(args).items()
                                if k not in ("enc_block", "dec_block", "twod_window_size", "spatial_pos", "use_vae",
                                             "defer_temporal_pool", "defer_spatial_pool", "gen_upscale")})
    path = tmp_path / "tok.ckpt"
    torch.save({"state_dict": sd, "hyper_parameters": {"args": old}}, path)
    m = OmniTokenizer_VQGAN.load_from_checkpoint(str(path), attention_mode="legacy")
    assert m.cfg.enc_block == "tttt" and m.cfg.window_size == 4 and m.cfg.spatial_pos == "rel"  # back-filled
    assert m.cfg.attention_mode == "legacy" and not m.use_vae
    got = m.state_dict()
    assert all(torch.equal(got[k], v) for k, v in sd.items() if k in got)
    from omnitokenizer_amd.vqgan import load_vqgan
    m2 = load_vqgan("omnitokenizer", str(path))
    assert not m2.training and m2.latent_shape == (4, 16, 16)  # (17 // 1, 64, 64) // (4, 4, 4)
    assert not any(k.startswith(("image_discriminator", "perceptual_model")) for k in got)


def test_torch_custom_ops_registered_with_schemas(lib):
    """the operators are schema-registered PyTorch custom ops (omnitok::*), with fake
    (shape-inference) kernels so that they trace without a GPU."""
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode
    from omnitokenizer_amd import ops
    assert ops.CUSTOM_OPS_REGISTERED
    assert str(torch.ops.omnitok.vq_argmin.default._schema) == "omnitok::vq_argmin(Tensor z, Tensor codebook) -> Tensor"
    for name in ("linear", "layernorm", "attn_spatial", "attn_window", "attn_temporal", "peg3d"):
        assert hasattr(torch.ops.omnitok, name)
    with FakeTensorMode():
        z, E = torch.empty(10, 3, 8, device="cuda"), torch.empty(8192, 8, device="cuda")
        ids = torch.ops.omnitok.vq_argmin(z, E)
        assert tuple(ids.shape) == (10, 3) and ids.dtype == torch.int64
        y = torch.ops.omnitok.linear(torch.empty(7, 512, device="cuda"), torch.empty(192, 512, device="cuda"), None)
        assert tuple(y.shape) == (7, 192)
        q = torch.empty(2048, 512, device="cuda")
        assert tuple(torch.ops.omnitok.attn_spatial(q, q, q, 1024, 8).shape) == (2048, 512)
        # the rest of the operator set
        s64 = torch.empty(64, device="cuda")
        assert tuple(torch.ops.omnitok.attn_spatial_h2(q, q, q, 1024, 8, s64, s64).shape) == (2048, 512)
        assert tuple(torch.ops.omnitok.linear_geglu(q, torch.empty(2816, 512, device="cuda")).shape) == (2048, 1408)
        vid = torch.empty(2, 3, 17, 256, 256, device="cuda")
        assert tuple(torch.ops.omnitok.patchify_ln(vid, 1, 4, 4, 8).shape) == (2 * 4 * 1024, 3 * 4 * 64)
        assert tuple(torch.ops.omnitok.pre_vq(q, torch.empty(8, 512, device="cuda"), torch.empty(8, device="cuda")).shape) \
            == (2048, 8)
        idl = torch.empty(2, 5, 32, 32, dtype=torch.int64, device="cuda")
        w, b = torch.empty(512, 8, device="cuda"), torch.empty(512, device="cuda")
        assert tuple(torch.ops.omnitok.dequant_post_vq(idl, torch.empty(8192, 8, device="cuda"), w, b).shape) == (2, 5, 32, 32, 512)
        assert tuple(torch.ops.omnitok.gather_rows(idl, torch.empty(8192, 512, device="cuda")).shape) == (2, 5, 32, 32, 512)
    for name in ("unpatchify", "qk_prep"):   # in-place operators: schema carries the mutation
        assert "!)" in str(getattr(torch.ops.omnitok, name).default._schema)


def test_vqgan_encode_decode_are_registered_operators():
    """OmniTokenizer_VQGAN.encode / decode run through omnitok::vqgan_encode / vqgan_decode; their fake
    implementations infer the shapes of every configuration family without a GPU."""
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode
    from omnitokenizer_amd import OmniTokenizer_VQGAN, make_args
    cases = [(dict(), (2, 3, 17, 256, 256), (2, 5, 32, 32)),
             (dict(enc_block="tawl", resolution=128), (1, 3, 5, 128, 128), (1, 2, 4, 4)),
             (dict(enc_block="ttnw", resolution=64), (1, 3, 5, 64, 64), (1, 2, 16, 16)),
             (dict(defer_spatial_pool=True, defer_temporal_pool=True, resolution=128), (1, 3, 9, 128, 128), (1, 3, 16, 16)),
             (dict(gen_upscale=2, resolution=64), (1, 3, 5, 64, 64), (1, 2, 8, 8))]
    for over, xs, ids_shape in cases:
        m = OmniTokenizer_VQGAN(make_args(2, **over))
        with FakeTensorMode():
            x = torch.empty(*xs, device="cuda")
            ids, emb, z = torch.ops.omnitok.vqgan_encode(x, m._handle, True, True)
            assert tuple(ids.shape) == ids_shape and ids.dtype == torch.int64
            assert tuple(emb.shape) == (ids_shape[0], 8) + ids_shape[1:] and tuple(z.shape) == ids_shape + (8,)
            pix = torch.ops.omnitok.vqgan_decode(ids, m._handle)
            # the decoder works from the latent grid: pooling / Up blocks of the encoder are not undone
            side = ids_shape[2] * m.cfg.dec_patch_size * (2 if over.get("defer_spatial_pool") else 1)
            assert tuple(pix.shape) == (xs[0], 3, xs[2], side, side), (over, pix.shape)
    with FakeTensorMode(), pytest.raises(RuntimeError, match="handle"):
        torch.ops.omnitok.vqgan_decode(torch.empty(1, 1, 8, 8, dtype=torch.int64, device="cuda"), 10 ** 9)
    with pytest.raises(NotImplementedError):   # CPU tensors: the operator exists for the GPU only (no fallback)
        torch.ops.omnitok.vqgan_decode(torch.empty(1, 1, 8, 8, dtype=torch.int64), 10 ** 9)


def test_lm_argument_errors_without_gpu(lib):
    """include/omnitok_lm.h: configuration / argument validation happens before any HIP call."""
    from omnitokenizer_amd._lib import OmnitokLmConfig
    h = ctypes.c_void_p()
    bad = OmnitokLmConfig(100, 16, 1, 3, 300)           # head_dim 100
    assert lib.omnitok_lm_create(ctypes.byref(bad), ctypes.byref(h)) != 0
    assert b"head_dim" in lib.omnitok_last_error()
    assert lib.omnitok_lm_create(None, ctypes.byref(h)) == -1
    ok = OmnitokLmConfig(8192, 5120, 24, 16, 1536)
    assert lib.omnitok_lm_create(ctypes.byref(ok), ctypes.byref(h)) == 0
    try:
        assert lib.omnitok_lm_step(h, None, None, None, 1, None, 1, None) == -1       # null pointers
        assert lib.omnitok_lm_alloc_cache(h, 17, 64) == -1                             # more than 16 streams
        assert lib.omnitok_lm_alloc_cache(h, 1, 9000) == -1                            # beyond 8192 positions
        assert lib.omnitok_lm_gemv(None, None, None, None, None, None, None, 1, 8, 256, 0, None) == -1
        shape = (ctypes.c_int64 * 2)(7, 7)
        x = ctypes.c_float(0)
        assert lib.omnitok_lm_set_weight(h, b"tok_emb.weight", ctypes.byref(x), shape, 2, None) == -1  # shape mismatch
        assert b"tok_emb.weight" in lib.omnitok_last_error()
        assert lib.omnitok_lm_set_weight(h, b"blocks.0.attn.mask", ctypes.byref(x), shape, 2, None) == 1  # ignored
    finally:
        lib.omnitok_lm_destroy(h)


def test_missing_library_fails_loudly(lib, monkeypatch):
    """No CPU / PyTorch fallback: without libomnitok.so the product raises instead of computing."""
    from omnitokenizer_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libomnitok.so")
    with pytest.raises(_lib.OmnitokError, match="no CPU fallback"):
        _lib.load()


def test_new_entry_points_validate_arguments_without_gpu(lib):
    """Argument validation of the round-2 entry points happens before any HIP call (status -1 + a message)."""
    import ctypes as C
    one = C.c_void_p(256)  # any non-null, 16-byte aligned value: the checks fire before it is dereferenced
    f = C.c_float
    # attn_pack: n_tokens not a multiple of 32; non-positive bounds
    assert lib.omnitok_attn_pack(one, 512, one, one, 1024, 48, 48, 8, None, None, one, one, f(8.0), f(8.0), f(1.0), f(1.0),
                                 None, 0, 0, one, one, one, None) == -1
    assert b"attn_pack" in lib.omnitok_last_error()
    assert lib.omnitok_attn_pack(one, 512, one, one, 1024, 64, 64, 8, None, None, one, one, f(8.0), f(0.0), f(1.0), f(1.0),
                                 None, 0, 0, one, one, one, None) == -1
    assert b"bounds" in lib.omnitok_last_error()
    # attn_spatial_h2: N not a multiple of 64
    assert lib.omnitok_attn_spatial_h2(one, one, one, one, 512, 1, 96, 8, f(8.0), f(1.0), f(1.0), None, 0, 0, None, 0, 0,
                                       None) == -1
    # lm_select: non-positive temperature, sampling without uniforms, top_p <= 0
    assert lib.omnitok_lm_select(one, None, 1, 100, f(0.0), f(1.0), f(0.0), -1, f(1.0), 0, None, one, None, None, None) == -1
    assert b"temperature" in lib.omnitok_last_error()
    assert lib.omnitok_lm_select(one, None, 1, 100, f(1.0), f(1.0), f(0.0), 10, f(0.9), 1, None, one, None, None, None) == -1
    assert lib.omnitok_lm_select(one, None, 1, 100, f(1.0), f(1.0), f(0.0), 10, f(0.0), 0, None, one, None, None, None) == -1
    # gather_rows / layernorm with the fused transpose: rows must be whole (a, c) groups; in-place is refused
    assert lib.omnitok_gather_rows_transposed(one, one, 10, one, 100, 3, 7, 512, None, None) == -1
    assert lib.omnitok_layernorm_transposed(one, one, None, one, 2, 3, 4, 512, f(1e-5), None) == -1
    # set_workspace: pointer / size mismatch and alignment are rejected on a null engine too
    assert lib.omnitok_engine_set_workspace(None, None, 0) == -1
    # plane GEMM: null argument block / null operands, the LayerNorm epilogue's full-row rule; per-engine options
    from omnitokenizer_amd._lib import OmnitokPlGemm
    assert lib.omnitok_gemm_pl(None, None) == -1
    g = OmnitokPlGemm()
    g.a = g.w = g.w_scale = g.c = g.out_planes = g.ln_gamma = 256
    g.M, g.N, g.K, g.ldc, g.epilogue, g.out_planes_k, g.out_bound = 64, 256, 512, 256, 2, 256, 30.0
    assert lib.omnitok_gemm_pl(C.byref(g), None) == -1 and b"512" in lib.omnitok_last_error()
    g.epilogue, g.N = 7, 512
    assert lib.omnitok_gemm_pl(C.byref(g), None) == -1
    assert lib.omnitok_pl_pack_weight(one, 512, 500, 512, 512, one, one, None) == -1   # N % 32
    assert lib.omnitok_pl_planes_bytes(1000, 512, 256) == 1024 * 512 * 4 and lib.omnitok_pl_planes_bytes(10, 500, 256) == -1
    assert lib.omnitok_engine_set_option(None, b"gemm_mode", 1) == -1
