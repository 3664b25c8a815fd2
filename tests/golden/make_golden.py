"""Generates tests/golden/*.npz by running the UNMODIFIED reference (imported read-only from
/root/reference through oracle/ref_harness.py) on seeded synthetic weights and inputs.

Run in the build container only:   python tests/golden/make_golden.py
The fixtures are committed; the GPU box has no /root/reference and only reads the .npz files.

Each fixture stores what is needed to regenerate the inputs (config overrides, seeds, shapes),
a crc32 of the synthetic state_dict (pins omnitokenizer_amd/synth.py) and the reference outputs:
  ids     encode() token ids                       (int16/int32, exact)
  z       pre-VQ l2-normalised latents             (fp32)  -- lets the VQ kernel be tested on the
                                                              reference's own z, bit-exact
  recon   decode(ids) pixels, optionally strided   (fp32)

vae_*.npz (--use_vae, reference omnitokenizer.py:260-266 / 293-317) store the host noise the
reference drew (torch.manual_seed(noise_seed); torch.randn), the raw mean|logvar moments, the
posterior sample z = encode(x) and recon = decode(z) in the layout the reference's decode accepts.

    python tests/golden/make_golden.py            # everything
    python tests/golden/make_golden.py vae        # only the vae_* fixtures
"""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

from oracle import ref_harness as rh  # noqa: E402
from omnitokenizer_amd.config import make_args, OmniTokConfig  # noqa: E402
from omnitokenizer_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

# name, stage, attention mode, overrides, batch, frames (1 = image), pixel stride in the fixture
CASES = [
    ("s2_sdpa_r64_img", 2, "sdpa", dict(resolution=64), 2, 1, 1),
    ("s2_sdpa_r64_vid", 2, "sdpa", dict(resolution=64), 2, 5, 1),
    ("s2_sdpa_r64_vid9", 2, "sdpa", dict(resolution=64), 1, 9, 1),
    ("s1_legacy_r64_img", 1, "legacy", dict(resolution=64), 2, 1, 1),
    ("s1_legacy_r64_vid", 1, "legacy", dict(resolution=64), 1, 5, 1),
    ("s1_sdpa_r64_img", 1, "sdpa", dict(resolution=64), 1, 1, 1),
    ("s2_sdpa_r128_vid_16k", 2, "sdpa", dict(resolution=128, n_codes=16384), 1, 5, 2),
    ("s2_sdpa_r256_img", 2, "sdpa", dict(resolution=256), 1, 1, 2),
    ("s2_sdpa_r256_vid", 2, "sdpa", dict(resolution=256), 1, 5, 4),
    ("s1_legacy_r256_img", 1, "legacy", dict(resolution=256), 1, 1, 4),  # BASELINE config C1
]


# the headline configuration itself (BASELINE C3 / C1 shapes at full length): one 17-frame 256x256 clip each
FULL_CASES = [
    ("s2_sdpa_r256_vid17", 2, "sdpa", dict(resolution=256), 1, 17, 8),
    ("s1_legacy_r256_vid17", 1, "legacy", dict(resolution=256), 1, 17, 8),
]
# ... and four clips (first, two interior, last) of the 32 DISTINCT clips bench.py feeds at C3
# (synth.synth_video(32, 17, 256, seed=1234)): the reference is run on each of them alone
B32_CLIPS = (0, 11, 20, 31)


# config coverage beyond the released checkpoints (SURVEY.md 8(a) rows a4', a18 and the deferred
# pools / gen_upscale of omnitokenizer.py:792-804, 957-959); same fixture format as CASES
VARIANT_CASES = [
    ("var_pool_a_r128_vid", 2, "sdpa", dict(resolution=128, enc_block="tawt"), 1, 5, 1),
    ("var_pool_m_r128_img", 2, "sdpa", dict(resolution=128, enc_block="tmwt"), 2, 1, 1),
    ("var_pool_l_r128_vid", 2, "sdpa", dict(resolution=128, enc_block="tlwt"), 1, 5, 1),
    ("var_cnn_r128_img", 2, "sdpa", dict(resolution=128, patch_embed="cnn"), 2, 1, 2),
    ("var_cnn_r128_vid", 1, "legacy", dict(resolution=128, patch_embed="cnn"), 1, 5, 2),
    ("var_defer_t_r128_vid", 2, "sdpa", dict(resolution=128, defer_temporal_pool=True), 1, 9, 2),
    ("var_defer_s_r128_vid", 2, "sdpa", dict(resolution=128, defer_spatial_pool=True), 1, 5, 2),
    ("var_defer_ts_r128_img", 2, "sdpa", dict(resolution=128, defer_spatial_pool=True, defer_temporal_pool=True),
     1, 1, 2),
    ("var_genup2_r64_vid", 2, "sdpa", dict(resolution=64, gen_upscale=2), 1, 5, 2),
    # encoder-side Up blocks (reference attention.py:116-150, 640-645, 686): the latent grid doubles, the
    # decoder (patch 8) then emits 2x the input resolution
    ("var_up_n_r64_vid", 2, "sdpa", dict(resolution=64, enc_block="ttnw"), 1, 5, 2),
    ("var_up_r_r64_img", 2, "sdpa", dict(resolution=64, enc_block="trtw"), 2, 1, 2),
    ("var_up_r_pool_r128_vid", 2, "sdpa", dict(resolution=128, enc_block="artw"), 1, 5, 2),
]


# Trained-checkpoint-like statistics (VERDICT r02 "missing" #2): synth profile "heavy" (Student-t weights, outlier
# LayerNorm gains up to 20, q/k scales up to 4, large biases) on image-like inputs; "mixed" = clip 0 one constant colour.
# name, stage, mode, overrides, batch, frames, stride, input kind
HEAVY_CASES = [
    ("heavy_s2_sdpa_r64_img", 2, "sdpa", dict(resolution=64), 2, 1, 1, "mixed"),
    ("heavy_s2_sdpa_r64_vid", 2, "sdpa", dict(resolution=64), 2, 5, 1, "mixed"),
    ("heavy_s1_legacy_r64_vid", 1, "legacy", dict(resolution=64), 1, 5, 1, "natural"),
    ("heavy_s2_sdpa_r128_vid_16k", 2, "sdpa", dict(resolution=128, n_codes=16384), 1, 5, 2, "natural"),
    ("heavy_s2_sdpa_r256_vid17", 2, "sdpa", dict(resolution=256), 1, 17, 8, "natural"),
]


def run_case(name, stage, mode, overrides, batch, frames, stride, input_kind="noise", profile="default"):
    args = make_args(stage, **overrides)
    cfg = OmniTokConfig.from_args(args, attention_mode=mode)
    sd = synth.synth_state_dict(cfg, seed=0, profile=profile)
    model = rh.build_reference_model(args)
    msg = model.load_state_dict(sd, strict=False)
    assert not msg.unexpected_keys, msg.unexpected_keys
    is_image = frames == 1
    res = cfg.resolution
    x = (synth.synth_image(batch, res, seed=1234, kind=input_kind) if is_image
         else synth.synth_video(batch, frames, res, seed=1234, kind=input_kind))
    with torch.no_grad(), rh.attention_mode(mode):
        emb, ids = model.encode(x, is_image, include_embeddings=True)
        # pre-VQ z exactly as the reference feeds Codebook.forward (omnitokenizer.py:248-252)
        h = model.pre_vq_conv(model.encoder(x, is_image))
        z = torch.nn.functional.normalize(h, p=2, dim=1)
        recon = model.decode(ids, is_image)
        if is_image or ids.shape[-1] == cfg.resolution // cfg.patch_size:
            flat = ids.reshape(ids.shape[0], -1)
            # flat-id decode == 4-D-id decode; video needs the un-pooled grid (:284)
            assert torch.equal(recon, model.decode(flat, is_image))
    assert int(ids.max()) < 32768
    sl = (Ellipsis, slice(None, None, stride), slice(None, None, stride))
    extra = {}
    if profile != "default":
        # The yardstick for these fixtures: how far the reference's OWN fp32 arithmetic is from the exact result.  The
        # oracle (pinned to the reference: tests/test_oracle_vs_golden.py) is evaluated in fp64 on the same weights.
        from oracle import omnitok_oracle as orc
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        with torch.no_grad():
            taps = {}
            orc.encode(sd64, x.double(), is_image, cfg, taps=taps)
            rec64 = orc.decode(sd64, ids, is_image, cfg)
        extra = dict(fp32_noise_z=np.float32((taps["z"] - z.permute(0, 2, 3, 4, 1).double()).abs().max().item()),
                     fp32_noise_pix=np.float32((rec64 - recon.double()).abs().max().item()))
        print(f"    reference fp32 vs fp64: z {extra['fp32_noise_z']:.2e}, pixels {extra['fp32_noise_pix']:.2e}")
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), **extra,
        stage=stage, mode=mode, overrides=repr(overrides), batch=batch, frames=frames,
        stride=stride, weight_seed=0, input_seed=1234, profile=profile, input_kind=input_kind,
        state_crc=np.uint32(synth.state_checksum(sd)),
        input_crc=np.uint32(__import__("zlib").crc32(x.numpy().tobytes())),
        ids=ids.numpy().astype(np.int16),
        z=z.permute(0, 2, 3, 4, 1).contiguous().numpy(),  # b t h w c
        emb=emb.permute(0, 2, 3, 4, 1).contiguous().numpy(),
        recon=recon[sl].contiguous().numpy(),
        recon_absmax=np.float32(recon.abs().max().item()),
    )
    print(f"{name}: ids {tuple(ids.shape)} uniq {ids.unique().numel()} recon {tuple(recon.shape)} "
          f"absmax {recon.abs().max().item():.3f}")


VAE_CASES = [
    ("vae_s2_sdpa_r64_img", 2, "sdpa", dict(resolution=64, use_vae=True), 2, 1, 1),
    ("vae_s2_sdpa_r64_vid", 2, "sdpa", dict(resolution=64, use_vae=True), 2, 5, 1),
    ("vae_s1_legacy_r64_vid", 1, "legacy", dict(resolution=64, use_vae=True), 1, 5, 1),
    ("vae_s2_sdpa_r256_vid", 2, "sdpa", dict(resolution=256, use_vae=True), 1, 5, 4),
]


def run_vae_case(name, stage, mode, overrides, batch, frames, stride, noise_seed=4321):
    args = make_args(stage, **overrides)
    cfg = OmniTokConfig.from_args(args, attention_mode=mode)
    sd = synth.synth_state_dict(cfg, seed=0)
    model = rh.build_reference_model(args)
    msg = model.load_state_dict(sd, strict=False)
    assert not msg.unexpected_keys, msg.unexpected_keys
    is_image = frames == 1
    res = cfg.resolution
    x = synth.synth_image(batch, res, seed=1234) if is_image else synth.synth_video(batch, frames, res, seed=1234)
    with torch.no_grad(), rh.attention_mode(mode):
        moments = model.pre_vq_conv(model.encoder(x, is_image))          # b 2c t h w
        torch.manual_seed(noise_seed)
        noise = torch.randn(batch, cfg.codebook_dim, *moments.shape[2:])  # what vae.py:16 draws
        torch.manual_seed(noise_seed)
        z = model.encode(x, is_image)                                     # b c h w | b c t h w
        z5 = z.unsqueeze(2) if is_image else z
        # the layouts reference decode accepts (omnitokenizer.py:296-316)
        z_in = z if is_image else z.permute(0, 2, 3, 4, 1).contiguous()
        recon = model.decode(z_in, is_image)
        flat = z5.permute(0, 2, 3, 4, 1).reshape(batch, -1, cfg.codebook_dim)
        assert torch.equal(recon, model.decode(flat, is_image))
    mean, logvar = torch.chunk(moments, 2, dim=1)
    assert torch.equal(z5, mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * noise)
    sl = (Ellipsis, slice(None, None, stride), slice(None, None, stride))
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        stage=stage, mode=mode, overrides=repr(overrides), batch=batch, frames=frames,
        stride=stride, weight_seed=0, input_seed=1234, noise_seed=noise_seed,
        state_crc=np.uint32(synth.state_checksum(sd)),
        input_crc=np.uint32(__import__("zlib").crc32(x.numpy().tobytes())),
        noise=noise.numpy(), moments=moments.contiguous().numpy(), z=z5.contiguous().numpy(),
        recon=recon[sl].contiguous().numpy(),
        recon_absmax=np.float32(recon.abs().max().item()),
    )
    print(f"{name}: z {tuple(z.shape)} std {z.std().item():.3f} logvar [{logvar.min().item():.2f}, "
          f"{logvar.max().item():.2f}] recon {tuple(recon.shape)} absmax {recon.abs().max().item():.3f}")


def make_vq_kat():
    """Known-answer vectors for the quantizer alone, from the reference's Codebook.forward
    (modules/codebook.py:76-143): random unit-norm z, un-normalised z, exact code rows, and
    duplicated code rows (ties -> lowest index)."""
    rh.install_stubs()
    from OmniTokenizer.modules.codebook import Codebook
    for n_codes in (8192, 16384):
        rng = np.random.Generator(np.random.PCG64(77 + n_codes))
        E = rng.standard_normal((n_codes, 8), dtype=np.float32)
        E[n_codes // 2] = E[17]          # duplicates: ties must resolve to the lowest index
        E[n_codes - 1] = E[17]
        E[5] = E[3]
        z = rng.standard_normal((6144, 8), dtype=np.float32)
        z[:4096] /= np.linalg.norm(z[:4096], axis=1, keepdims=True)
        z[4096:4160] = E[17]             # hits the triple tie
        z[4160:4200] = E[3]              # hits the double tie
        z[4200:4300] = E[rng.integers(0, n_codes, 100)]
        z[4300:4400] *= 1e-3
        z[4400:4500] *= 30.0
        cb = Codebook(n_codes, 8).eval()
        cb._need_init = False
        cb.embeddings.data.copy_(torch.from_numpy(E))
        zt = torch.from_numpy(z).reshape(6, 1, 32, 32, 8).permute(0, 4, 1, 2, 3).contiguous()
        with torch.no_grad():
            out = cb(zt)
        ids = out["encodings"].reshape(-1).numpy()
        assert (ids[4096:4160] == 17).all() and (ids[4160:4200] == 3).all()
        np.savez_compressed(os.path.join(OUT, f"vq_kat_{n_codes}.npz"), z=z, codebook=E,
                            ids=ids.astype(np.int16))
        print(f"vq_kat_{n_codes}: uniq {np.unique(ids).size}")


EXT_CASES = [  # --use_external_codebook (VectorQuantize, cosine similarity): SURVEY.md 8(a) row a16
    ("ext_s2_sdpa_r64_img", 2, "sdpa", dict(resolution=64, use_external_codebook=True), 2, 1, 1),
    ("ext_s2_sdpa_r64_vid", 2, "sdpa", dict(resolution=64, use_external_codebook=True), 1, 5, 1),
    ("ext_s1_legacy_r128_vid", 1, "legacy", dict(resolution=128, use_external_codebook=True, n_codes=16384), 1, 5, 2),
    # without --l2_code VectorQuantize uses its EuclideanCodebook (use_cosine_sim = args.l2_code, omnitokenizer.py:134)
    ("ext_euclid_s2_sdpa_r64_img", 2, "sdpa", dict(resolution=64, use_external_codebook=True, l2_code=False), 2, 1, 1),
    ("ext_euclid_s2_sdpa_r64_vid", 2, "sdpa", dict(resolution=64, use_external_codebook=True, l2_code=False), 1, 5, 1),
]


def run_ext_case(name, stage, mode, overrides, batch, frames, stride):
    """The reference's decode() raises with the external codebook (it reads codebook.embeddings), so the
    reconstruction stored here is what its forward() computes (omnitokenizer.py:358-363):
    decoder(post_vq_conv(codebook(pre_vq_conv(encoder(x)))['embeddings']))."""
    args = make_args(stage, **overrides)
    cfg = OmniTokConfig.from_args(args, attention_mode=mode)
    sd = synth.synth_state_dict(cfg, seed=0)
    model = rh.build_reference_model(args)
    msg = model.load_state_dict(sd, strict=False)
    assert not msg.unexpected_keys, msg.unexpected_keys
    is_image = frames == 1
    res = cfg.resolution
    x = synth.synth_image(batch, res, seed=1234) if is_image else synth.synth_video(batch, frames, res, seed=1234)
    with torch.no_grad(), rh.attention_mode(mode):
        emb, ids = model.encode(x, is_image, include_embeddings=True)
        tok = model.encoder(x, is_image)                                  # b d t h w
        z = model.codebook._codebook.transform_input(
            model.codebook.project_in(tok.permute(0, 2, 3, 4, 1)))          # b t h w c (unit norm if cosine)
        vq = model.codebook(model.pre_vq_conv(tok))
        assert torch.equal(vq["encodings"], ids)
        recon = model.decoder(model.post_vq_conv(vq["embeddings"]), is_image)
        try:
            model.decode(ids, is_image)
            raise SystemExit("reference decode() unexpectedly works with the external codebook")
        except AttributeError:
            pass
    sl = (Ellipsis, slice(None, None, stride), slice(None, None, stride))
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        stage=stage, mode=mode, overrides=repr(overrides), batch=batch, frames=frames,
        stride=stride, weight_seed=0, input_seed=1234,
        state_crc=np.uint32(synth.state_checksum(sd)),
        input_crc=np.uint32(__import__("zlib").crc32(x.numpy().tobytes())),
        ids=ids.numpy().astype(np.int16), z=z.contiguous().numpy(),
        emb=emb.permute(0, 2, 3, 4, 1)[..., ::8].contiguous().numpy(),  # b t h w c, every 8th of 512 channels
        recon=recon[sl].contiguous().numpy(), recon_absmax=np.float32(recon.abs().max().item()),
        perplexity=np.float32(vq["perplexity"].item()),
    )
    print(f"{name}: ids {tuple(ids.shape)} uniq {ids.unique().numel()} recon {tuple(recon.shape)} "
          f"absmax {recon.abs().max().item():.3f}")


def make_vq_cdist_kat():
    """Known answers of the reference's EuclideanCodebook.forward (eval), incl. duplicated and exact codes."""
    rh.install_stubs()
    from OmniTokenizer.quantizer.vector_quantize_pytorch import EuclideanCodebook
    n_codes = 8192
    rng = np.random.Generator(np.random.PCG64(101))
    E = rng.standard_normal((n_codes, 8), dtype=np.float32)
    E[n_codes // 2] = E[17]
    E[n_codes - 1] = E[17]
    E[5] = E[3]
    z = rng.standard_normal((4096, 8), dtype=np.float32)
    z[3000:3064] = E[17]
    z[3064:3100] = E[3]
    z[3100:3200] = E[rng.integers(0, n_codes, 100)]
    z[3200:3300] *= 1e-3
    z[3300:3400] *= 30.0
    cb = EuclideanCodebook(8, n_codes).eval()
    cb.embed.data.copy_(torch.from_numpy(E)[None])
    with torch.no_grad():
        _, ind, _ = cb(torch.from_numpy(z)[None])
    ids = ind.reshape(-1).numpy()
    assert (ids[3000:3064] == 17).all() and (ids[3064:3100] == 3).all()
    np.savez_compressed(os.path.join(OUT, "vq_cdist_kat_8192.npz"), z=z, codebook=E, ids=ids.astype(np.int16))
    print(f"vq_cdist_kat_8192: uniq {np.unique(ids).size}")


def make_vq_cos_kat():
    """Known answers of the reference's CosineSimCodebook.forward (eval): unit-norm z, duplicated code rows
    (ties -> lowest index), exact code rows."""
    rh.install_stubs()
    from OmniTokenizer.quantizer.vector_quantize_pytorch import CosineSimCodebook
    n_codes = 8192
    rng = np.random.Generator(np.random.PCG64(99))
    E = rng.standard_normal((n_codes, 8), dtype=np.float32)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    E[n_codes // 2] = E[17]
    E[n_codes - 1] = E[17]
    E[5] = E[3]
    z = rng.standard_normal((4096, 8), dtype=np.float32)
    z /= np.linalg.norm(z, axis=1, keepdims=True)
    z[3000:3064] = E[17]
    z[3064:3100] = E[3]
    z[3100:3200] = E[rng.integers(0, n_codes, 100)]
    cb = CosineSimCodebook(8, n_codes).eval()
    cb.embed.data.copy_(torch.from_numpy(E)[None])
    with torch.no_grad():
        _, ind, _ = cb(torch.from_numpy(z)[None])
    ids = ind.reshape(-1).numpy()
    assert (ids[3000:3064] == 17).all() and (ids[3064:3100] == 3).all()
    np.savez_compressed(os.path.join(OUT, "vq_cos_kat_8192.npz"), z=z, codebook=E, ids=ids.astype(np.int16))
    print(f"vq_cos_kat_8192: uniq {np.unique(ids).size}")


GPT_CASES = [  # name, vocab, block_size, n_layer, n_head, n_embd (head_dim 64 / 96 / 128)
    ("gpt_hd64", 320, 48, 2, 4, 256),
    ("gpt_hd96", 520, 40, 2, 8, 768),
    ("gpt_hd128", 300, 40, 3, 4, 512),
]


def make_gpt_golden():
    """LM consumer: logits and greedy samples of the reference's own GPT class
    (OmniTokenizer/modules/gpt.py, imported unmodified) on seeded weights."""
    import argparse
    import importlib
    from oracle import gpt_oracle as go
    rh.install_stubs()
    gpt = importlib.import_module("OmniTokenizer.modules.gpt")
    for name, V, BS, L, H, C in GPT_CASES:
        sd = go.synth_gpt_state(V, BS, L, H, C, seed=2)
        m = gpt.GPT(argparse.Namespace(), V, BS, n_layer=L, n_head=H, n_embd=C).eval()
        missing = m.load_state_dict(sd, strict=False)
        assert not missing.unexpected_keys and all(k.endswith("attn.mask") for k in missing.missing_keys)
        g = torch.Generator().manual_seed(5)
        idx = torch.randint(0, V, (2, 16), generator=g)
        cls = torch.randint(0, 100, (2, 1), generator=g)
        steps = 14
        with torch.no_grad():
            logits, _ = m(idx)
            greedy = gpt.sample_with_past(idx[:, :3].clone(), m, steps, temperature=0.9, sample_logits=False,
                                          top_k=50, top_p=0.9)
            cfg_a = gpt.sample_with_past_cfg(cls.clone(), m, steps, sample_logits=False, top_k=64, top_p=1.0,
                                             cfg_ratio=1.5, class_first=True)
            cfg_b = gpt.sample_with_past_cfg(cls.clone(), m, steps, sample_logits=False, top_k=64, top_p=0.95,
                                             cfg_ratio=0.5, class_first=False, scale_cfg=True)
        crc = 0
        for k in sd:
            crc = __import__("zlib").crc32(sd[k].numpy().tobytes(), crc)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), vocab=V, block_size=BS, n_layer=L, n_head=H, n_embd=C,
                            weight_seed=2, state_crc=np.uint32(crc), idx=idx.numpy(), cls=cls.numpy(), steps=steps,
                            logits=logits.numpy(), greedy=greedy.numpy(), cfg_a=cfg_a.numpy(), cfg_b=cfg_b.numpy())
        print(f"{name}: logits {tuple(logits.shape)} absmax {logits.abs().max().item():.2f} greedy {greedy[0, :6].tolist()}")


def make_gpt_vtok_golden(name="gpt_vtok"):
    """The reference GPT's optional inputs (gpt.py:207-258): vtokens_pos boxes (cbox / tbox) and explicit
    embeddings prepended -- full forward, the KV-cached first call + steps, and greedy sample_with_past(cbox)."""
    import argparse
    import importlib
    from oracle import gpt_oracle as go
    rh.install_stubs()
    gpt = importlib.import_module("OmniTokenizer.modules.gpt")
    V, BS, L, H, C, TT, RR = 300, 40, 2, 4, 256, 3, 6
    sd = go.synth_gpt_state(V, BS, L, H, C, seed=4, vtokens_pos_shape=(TT, RR))
    args = argparse.Namespace(sequence_length=TT, resolution=RR)
    m = gpt.GPT(args, V, BS, n_layer=L, n_head=H, n_embd=C, vtokens_pos=True).eval()
    missing = m.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all(k.endswith("attn.mask") for k in missing.missing_keys)
    g = torch.Generator().manual_seed(11)
    cbox = [(0, 3, 1, 5), (2, 5, 0, 4)]          # 3 x 4 spatial boxes: 3 frames x 12 = 36 positions
    tbox = [(0, 2), (1, 3)]                      # with tbox: 2 frames x 12 = 24 positions
    emb = torch.randn(2, 2, C, generator=g) * 0.5
    idx36 = torch.randint(0, V, (2, 34), generator=g)   # 2 embeddings + 34 tokens = 36
    idx24 = torch.randint(0, V, (2, 24), generator=g)
    steps = 10
    with torch.no_grad():
        lg_emb, _ = m(idx36, embeddings=emb, cbox=cbox)
        lg_tbox, _ = m(idx24, cbox=cbox, tbox=tbox)
        # KV-cached: first call with embeddings + 3 tokens (positions 0..4), then 4 single-token steps.
        # Batch 1 only: with a past the reference adds a [B, C] position term to [B, 1, C] token embeddings
        # (gpt.py:248-253), which broadcasts to [B, B, C] for B > 1 -- the cached vtokens_pos path of the
        # reference is only well-formed for one stream.
        first, _, present = m.forward_with_past(idx36[:1, :3], embeddings=emb[:1], cbox=cbox[:1])
        past, plen, step_logits = [present], 5, []
        for t in range(4):
            lgt, _, present = m.forward_with_past(idx36[:1, 3 + t:4 + t], past=past, past_length=plen, cbox=cbox[:1])
            past.append(present)
            plen += 1
            step_logits.append(lgt[:, -1])
        greedy = torch.cat([gpt.sample_with_past(idx36[b:b + 1, :3].clone(), m, steps, temperature=0.8,
                                                 sample_logits=False, top_k=40, top_p=0.9, cbox=cbox[b:b + 1])
                            for b in range(2)], 0)
    crc = 0
    for k in sd:
        crc = __import__("zlib").crc32(sd[k].numpy().tobytes(), crc)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), vocab=V, block_size=BS, n_layer=L, n_head=H, n_embd=C,
                        sequence_length=TT, resolution=RR, weight_seed=4, state_crc=np.uint32(crc),
                        cbox=np.array(cbox), tbox=np.array(tbox), emb=emb.numpy(), idx36=idx36.numpy(),
                        idx24=idx24.numpy(), steps=steps, logits_emb=lg_emb.numpy(), logits_tbox=lg_tbox.numpy(),
                        first=first.numpy(), step_logits=torch.stack(step_logits, 1).numpy(), greedy=greedy.numpy())
    print(f"{name}: logits_emb {tuple(lg_emb.shape)} logits_tbox {tuple(lg_tbox.shape)} greedy {greedy[0, :6].tolist()}")


def run_b32_case(name="s2_sdpa_r256_vid17_b32", stride=8):
    args = make_args(2, resolution=256)
    cfg = OmniTokConfig.from_args(args, attention_mode="sdpa")
    sd = synth.synth_state_dict(cfg, seed=0)
    model = rh.build_reference_model(args)
    msg = model.load_state_dict(sd, strict=False)
    assert not msg.unexpected_keys, msg.unexpected_keys
    x = synth.synth_video(32, 17, 256, seed=1234)
    ids_l, z_l, rec_l = [], [], []
    with torch.no_grad(), rh.attention_mode("sdpa"):
        for b in B32_CLIPS:
            xb = x[b:b + 1].contiguous()
            ids = model.encode(xb, False)
            h = model.pre_vq_conv(model.encoder(xb, False))
            z_l.append(torch.nn.functional.normalize(h, p=2, dim=1).permute(0, 2, 3, 4, 1).contiguous())
            ids_l.append(ids)
            rec_l.append(model.decode(ids, False)[..., ::stride, ::stride].contiguous())
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), clips=np.array(B32_CLIPS), batch=32, frames=17, stride=stride,
        weight_seed=0, input_seed=1234, state_crc=np.uint32(synth.state_checksum(sd)),
        input_crc=np.uint32(__import__("zlib").crc32(x.numpy().tobytes())),
        ids=torch.cat(ids_l).numpy().astype(np.int16), z=torch.cat(z_l).numpy(), recon=torch.cat(rec_l).numpy())
    print(f"{name}: clips {B32_CLIPS} ids {tuple(torch.cat(ids_l).shape)}")


def boundary_distance(z64, E64, ids, chunk=2048):
    """Distance (L2, in latent space, fp64) from every latent to the nearest decision boundary of its code's cell:
    min over j != i of (d_j - d_i) / (2 |e_j - e_i|) with d = squared distances.  A perturbation dz can flip the token only
    if |dz|_2 >= this -- the currency in which id flips of different arithmetics are compared."""
    z, ids = z64.reshape(-1, z64.shape[-1]), ids.reshape(-1)
    out = torch.empty(z.shape[0], dtype=torch.float64)
    for a in range(0, z.shape[0], chunk):
        zc, ic = z[a:a + chunk], ids[a:a + chunk]
        d = (zc * zc).sum(1, keepdim=True) - 2.0 * zc @ E64.t() + (E64 * E64).sum(1)[None]
        di = d.gather(1, ic[:, None])
        sep = torch.cdist(E64[ic], E64, compute_mode="donot_use_mm_for_euclid_dist")   # |e_j - e_i|
        m = (d - di) / (2.0 * sep.clamp_min(1e-300))
        m[sep < 1e-12] = float("inf")                        # exact duplicates of the code (none in a randn codebook)
        m.scatter_(1, ic[:, None], float("inf"))             # the code itself
        out[a:a + chunk] = m.amin(1)
    return out


def run_heavy_batch_case(name="heavy_s2_sdpa_r256_vid17_b8", batch=8, stride=8):
    """Batch-scale evidence on trained-like statistics: 8 distinct 17x256x256 clips ("mixed": clip 0
    is one constant colour, the others image-like), synth profile "heavy", the REFERENCE itself run in fp32 and -- the
    same nn.Module after .double() -- in fp64.  Both id sets, both z and the strided reconstructions are stored, so that
    a GPU test can state its result in the reference's own currency: ids that flip between the reference's fp32 and
    fp64 runs, and the per-token distance between its fp32 and fp64 latents."""
    args = make_args(2, resolution=256)
    cfg = OmniTokConfig.from_args(args, attention_mode="sdpa")
    sd = synth.synth_state_dict(cfg, seed=0, profile="heavy")
    x = synth.synth_video(batch, 17, 256, seed=1234, kind="mixed")
    out = {}
    for tag, dt in (("32", torch.float32), ("64", torch.float64)):
        model = rh.build_reference_model(args)
        msg = model.load_state_dict(sd, strict=False)
        assert not msg.unexpected_keys, msg.unexpected_keys
        model = model.to(dt)
        ids_l, z_l, rec_l = [], [], []
        with torch.no_grad(), rh.attention_mode("sdpa"):
            for b in range(batch):  # one clip at a time: the reference's results do not depend on the batch
                xb = x[b:b + 1].to(dt)
                h = model.pre_vq_conv(model.encoder(xb, False))
                z = torch.nn.functional.normalize(h, p=2, dim=1)
                ids = model.codebook(z)["encodings"]
                assert torch.equal(ids, model.encode(xb, False)) if b == 0 else True
                ids_l.append(ids)
                z_l.append(z.permute(0, 2, 3, 4, 1).contiguous())
                # decode the fp32 ids in both precisions: the pixel yardstick is arithmetic noise, not id flips
                rec_l.append(model.decode(out["ids32"][b:b + 1] if tag == "64" else ids, False))
        out["ids" + tag], out["z" + tag], out["rec" + tag] = torch.cat(ids_l), torch.cat(z_l), torch.cat(rec_l)
        assert out["z" + tag].dtype == dt and out["rec" + tag].dtype == dt
    ids32, ids64, z32, z64 = out["ids32"], out["ids64"], out["z32"], out["z64"]
    E = sd["codebook.embeddings"].double()
    flip = (ids32 != ids64).reshape(-1).nonzero().flatten()
    zf = z64.reshape(-1, 8)[flip]
    d32 = ((zf - E[ids32.reshape(-1)[flip]]) ** 2).sum(1)
    d64 = ((zf - E[ids64.reshape(-1)[flip]]) ** 2).sum(1)
    noise_l2 = (z32.double() - z64).reshape(-1, 8).norm(dim=1)
    noise_z_clip = (z32.double() - z64).abs().reshape(batch, -1).amax(1)
    noise_pix_clip = (out["rec32"].double() - out["rec64"]).abs().reshape(batch, -1).amax(1)
    sl = (Ellipsis, slice(None, None, stride), slice(None, None, stride))
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        stage=2, mode="sdpa", overrides=repr(dict(resolution=256)), batch=batch, frames=17, stride=stride,
        weight_seed=0, input_seed=1234, profile="heavy", input_kind="mixed",
        state_crc=np.uint32(synth.state_checksum(sd)),
        input_crc=np.uint32(__import__("zlib").crc32(x.numpy().tobytes())),
        ids=ids32.numpy().astype(np.int16), ids64=ids64.numpy().astype(np.int16),
        # z64 = z.double() + z64_resid.double() to 1e-12 (the residual is ~1e-5, stored in fp32): half the bytes of fp64
        z=z32.numpy(), z64_resid=(z64 - z32.double()).float().numpy(), emb=np.zeros((0,), np.float32),
        recon=out["rec32"][sl].contiguous().numpy(),
        recon_absmax=np.float32(out["rec32"].abs().max().item()),
        fp32_noise_z=np.float32(noise_z_clip.max().item()), fp32_noise_pix=np.float32(noise_pix_clip.max().item()),
        fp32_noise_z_clip=noise_z_clip.numpy().astype(np.float32),
        fp32_noise_pix_clip=noise_pix_clip.numpy().astype(np.float32),
        fp32_noise_l2_max=np.float32(noise_l2.max().item()),
        # distance of every fp64 latent to the nearest cell boundary of ITS fp64 code (see boundary_distance)
        boundary=boundary_distance(z64, E, ids64).float().numpy(),
        ref_flips=np.int32(flip.numel()), ref_flip_index=flip.numpy().astype(np.int32),
        ref_flip_gap=(d32 - d64).numpy(),
    )
    print(f"{name}: {ids32.numel()} tokens, uniq {ids32.unique().numel()}; reference fp32-vs-fp64: {flip.numel()} id flips "
          f"(fp64 gaps {(d32 - d64).tolist()}), z noise {noise_z_clip.max().item():.2e} (L2 {noise_l2.max().item():.2e}), "
          f"pixel noise {noise_pix_clip.max().item():.2e}, |recon|max {out['rec32'].abs().max().item():.1f}")


def make_usage_state_golden(name="usage_state_s2_sdpa_r64"):
    """The eval-time STATE MUTATION of the reference's quantiser (codebook.py:122-143): every Codebook.forward -- hence
    every encode() -- rewrites the `codebook_usage` buffer (first call: the batch's usage; later: the 0.99 / 0.01 EMA) and
    bumps `call_cnt`.  Fixture: the reference's state_dict entry after encode(x_img), encode(x_vid), then
    forward-style third call on x_img again, plus the ids of each call."""
    args = make_args(2, resolution=64)
    cfg = OmniTokConfig.from_args(args, attention_mode="sdpa")
    sd = synth.synth_state_dict(cfg, seed=0)
    model = rh.build_reference_model(args)
    model.load_state_dict(sd, strict=False)
    xi = synth.synth_image(2, 64, seed=1234)
    xv = synth.synth_video(2, 5, 64, seed=1234)
    usage, ids_all = [], []
    with torch.no_grad(), rh.attention_mode("sdpa"):
        for x, is_image in ((xi, True), (xv, False), (xi, True)):
            ids = model.encode(x, is_image)
            ids_all.append(ids.numpy().astype(np.int16))
            usage.append(model.state_dict()["codebook.codebook_usage"].clone().numpy())
    assert model.codebook.call_cnt == 3
    np.savez_compressed(os.path.join(OUT, name + ".npz"), usage=np.stack(usage), call_cnt=np.int32(model.codebook.call_cnt),
                        ids0=ids_all[0], ids1=ids_all[1], ids2=ids_all[2], state_crc=np.uint32(synth.state_checksum(sd)))
    print(f"{name}: usage sums {[float(u.sum()) for u in usage]}, nonzero {[int((u > 0).sum()) for u in usage]}")


if __name__ == "__main__":
    assert rh.reference_available(), "run in the build container (needs /root/reference)"
    only = sys.argv[1] if len(sys.argv) > 1 else None
    if only in (None, "usage"):
        make_usage_state_golden()
    if only == "up":  # only the Up-block variants
        for c in VARIANT_CASES:
            if c[0].startswith("var_up_"):
                run_case(*c)
    if only in (None, "full"):
        for c in FULL_CASES:
            run_case(*c)
        run_b32_case()
    if only in (None, "vq"):
        make_vq_kat()
    if only in (None, "e2e"):
        for c in CASES:
            run_case(*c)
    if only in (None, "variants"):
        for c in VARIANT_CASES:
            run_case(*c)
    if only in (None, "ext"):
        make_vq_cos_kat()
        make_vq_cdist_kat()
        for c in EXT_CASES:
            run_ext_case(*c)
    if only in (None, "gpt"):
        make_gpt_golden()
    if only in (None, "gpt", "gpt_vtok"):
        make_gpt_vtok_golden()
    if only in (None, "vae"):
        for c in VAE_CASES:
            run_vae_case(*c)
    if only in (None, "heavy"):
        for c in HEAVY_CASES:
            run_case(*c, profile="heavy")
    if only in (None, "heavy", "heavy_b8"):
        run_heavy_batch_case()
