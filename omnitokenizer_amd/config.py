"""Typed configuration for the encode/decode path, parsed from the reference's argparse
Namespace.

The reference constructs `OmniTokenizer_VQGAN(args)` from three stacked argparse groups
(reference OmniTokenizer/base.py:245-269, omnitokenizer.py:695-768, data.py:552-577) and
back-fills attributes missing from old checkpoints with hasattr() (omnitokenizer.py:70-98,
121-125).  `OmniTokConfig.from_args` applies the same defaults; `make_args` builds the
Namespace the released scripts produce (scripts/recons/train.sh:2-13 = "stage1",
scripts/recons/eval_image_inet.sh:1-9 / eval_video.sh:1-9 = "stage2").
"""
from __future__ import annotations

import argparse
from dataclasses import dataclass


def _get(args, name, default):
    return getattr(args, name) if hasattr(args, name) else default


@dataclass(frozen=True)
class OmniTokConfig:
    resolution: int
    sequence_length: int
    image_channels: int
    patch_size: int
    temporal_patch_size: int
    patch_embed: str
    enc_block: str
    dec_block: str
    window_size: int
    spatial_pos: str
    spatial_depth: int
    temporal_depth: int
    causal_in_temporal_transformer: bool
    causal_in_peg: bool
    dim: int
    dim_head: int
    heads: int
    ff_mult: float
    n_codes: int
    codebook_dim: int
    l2_code: bool
    use_vae: bool
    use_external_codebook: bool
    attention_mode: str  # "sdpa" | "legacy"  (reference attention.py:439 picks by torch version)
    defer_temporal_pool: bool = False  # omnitokenizer.py:792-797 / 985-990 ('linear' patch-embed only)
    defer_spatial_pool: bool = False   # omnitokenizer.py:799-804 / 992-1003
    gen_upscale: int = 1               # decoder patch_size *= gen_upscale, omnitokenizer.py:957-959
    norm_type: str = "batch"           # Normalize() of the 'cnn' patch-embed, base.py:272-277
    codebook_type: str = "vq"          # --codebook_type (only 'vq' exists for the external quantiser)

    @property
    def ff_inner(self) -> int:
        # reference attention.py:161  inner_dim = int(mult * (2 / 3) * dim)
        return int(self.ff_mult * (2 / 3) * self.dim)

    # patch sizes actually used (the deferred pools only exist for patch_embed == 'linear')
    @property
    def _defer_t(self) -> bool:
        return self.defer_temporal_pool and self.patch_embed == "linear"

    @property
    def _defer_s(self) -> bool:
        return self.defer_spatial_pool and self.patch_embed == "linear"

    @property
    def enc_patch_size(self) -> int:
        return self.patch_size // 2 if self._defer_s else self.patch_size

    @property
    def enc_temporal_patch_size(self) -> int:
        return self.temporal_patch_size // 2 if self._defer_t else self.temporal_patch_size

    @property
    def dec_patch_size(self) -> int:
        p = self.patch_size * self.gen_upscale
        return p // 2 if self._defer_s else p

    @property
    def dec_temporal_patch_size(self) -> int:
        return self.enc_temporal_patch_size

    @property
    def enc_grid_divisor(self) -> int:
        """latent grid = (resolution / enc_patch_size) / enc_grid_divisor."""
        n_pool = sum(self.enc_block.count(c) for c in "aml")
        return (2 ** n_pool) * (2 if self._defer_s else 1)

    @property
    def enc_grid_multiplier(self) -> int:
        """Up blocks ('n' / 'r', reference attention.py:116-150) in the encoder double the grid:
        latent grid = (resolution / enc_patch_size) * enc_grid_multiplier / enc_grid_divisor."""
        return 2 ** sum(self.enc_block.count(c) for c in "nr")

    @staticmethod
    def from_args(args, attention_mode: str = "sdpa") -> "OmniTokConfig":
        spatial_depth = args.spatial_depth
        enc_block = _get(args, "enc_block", "t" * spatial_depth)  # omnitokenizer.py:70-74
        dec_block = _get(args, "dec_block", "t" * spatial_depth)
        cfg = OmniTokConfig(
            resolution=int(args.resolution),
            sequence_length=int(_get(args, "sequence_length", 17)),
            image_channels=int(_get(args, "image_channels", 3)),
            patch_size=int(args.patch_size),
            temporal_patch_size=int(args.temporal_patch_size),
            patch_embed=str(args.patch_embed),
            enc_block=str(enc_block),
            dec_block=str(dec_block),
            window_size=int(_get(args, "twod_window_size", 4)),  # omnitokenizer.py:76-77
            spatial_pos=str(_get(args, "spatial_pos", "rel")),  # omnitokenizer.py:83-84
            spatial_depth=int(spatial_depth),
            temporal_depth=int(args.temporal_depth),
            causal_in_temporal_transformer=bool(args.causal_in_temporal_transformer),
            causal_in_peg=bool(args.causal_in_peg),
            dim=int(args.embedding_dim),
            dim_head=int(args.dim_head),
            heads=int(args.heads),
            ff_mult=float(args.ff_mult),
            n_codes=int(args.n_codes),
            codebook_dim=int(args.codebook_dim),
            l2_code=bool(args.l2_code),
            use_vae=bool(_get(args, "use_vae", False)),
            use_external_codebook=bool(_get(args, "use_external_codebook", False)),
            attention_mode=attention_mode,
            defer_temporal_pool=bool(_get(args, "defer_temporal_pool", False)),  # omnitokenizer.py:79-81
            defer_spatial_pool=bool(_get(args, "defer_spatial_pool", False)),
            gen_upscale=int(_get(args, "gen_upscale", None) or 1),                # omnitokenizer.py:91-92
            norm_type=str(_get(args, "norm_type", "batch")),
            codebook_type=str(_get(args, "codebook_type", "vq")),
        )
        cfg.validate()
        return cfg

    def validate(self):
        """Reject configurations outside the built path with a clear error instead of
        computing something else (SURVEY.md section 8 a18/a19/a16/a4')."""
        if self.attention_mode not in ("sdpa", "legacy"):
            raise ValueError(f"attention_mode must be 'sdpa' or 'legacy', got {self.attention_mode!r}")
        if self.patch_embed not in ("linear", "cnn"):
            raise NotImplementedError(f"patch_embed={self.patch_embed!r} (reference omnitokenizer.py:839-840)")
        if self.patch_embed == "cnn" and self.norm_type != "batch":
            # GroupNorm(32 groups) over the decoder's 3 output channels cannot be constructed
            raise ValueError("patch_embed='cnn' needs norm_type='batch': the reference's "
                             "Normalize(image_channel, 'group') raises 'num_channels (3) must be divisible by "
                             "num_groups (32)' (base.py:274, omnitokenizer.py:1023)")
        bad = sorted(set(self.enc_block) - set("twamlnr"))
        if bad:
            raise NotImplementedError(f"enc_block={self.enc_block!r}: unknown block types {bad} "
                                      "(reference attention.py:614-649)")
        bad = sorted(set(self.dec_block) - set("tw"))
        if bad:
            raise NotImplementedError(
                f"dec_block={self.dec_block!r}: block types {bad} are not built. In the DECODER 'n'/'r' (Up) blocks "
                "make the reference itself raise (einops shape mismatch at omnitokenizer.py:1078: it regroups "
                "(b h w) rows with h // down_ratio); in the encoder they are built. Pooling blocks are "
                "encoder-side")
        if self.gen_upscale < 1:
            raise ValueError("gen_upscale must be >= 1")
        if (self._defer_s and self.patch_size % 2) or (self._defer_t and self.temporal_patch_size % 2):
            raise ValueError("deferred pooling needs an even patch size")
        if len(self.enc_block) != self.spatial_depth:
            raise ValueError("len(enc_block) must equal spatial_depth (reference attention.py:608)")
        if self.use_external_codebook:
            # VectorQuantize (quantizer/vector_quantize_pytorch.py:690), built by omnitokenizer.py:131-138
            if self.codebook_type != "vq":
                raise NotImplementedError(f"codebook_type={self.codebook_type!r} (reference omnitokenizer.py:139-140)")
            if self.use_vae:
                raise NotImplementedError("use_vae with use_external_codebook: pre_vq_conv is Identity there "
                                          "(omnitokenizer.py:136), the posterior would have 512 channels")
        if self.dim != self.dim_head * self.heads or self.dim_head != 64:
            raise NotImplementedError("kernels are built for dim == heads*dim_head with dim_head == 64")
        if self.dim % 128 != 0:
            raise NotImplementedError("embedding_dim must be a multiple of 128")
        if self.spatial_pos not in ("rel", "rope"):
            raise ValueError(f"spatial_pos={self.spatial_pos!r}")
        if self.resolution % self.enc_patch_size:
            raise ValueError("resolution must be divisible by patch_size (reference omnitokenizer.py:789-790)")


def make_args(stage: int = 2, **overrides) -> argparse.Namespace:
    """Namespace equal to what the released scripts pass to the reference constructor.

    stage=1: scripts/recons/train.sh:2-13 (pt=2, spatial_pos default "rel") -- imagenet_only.
    stage=2: scripts/recons/eval_image_inet.sh:1-9 (pt=4, rope)             -- imagenet_k600 etc.
    Training-only attributes the reference constructor reads are filled with the argparse
    defaults so that the same Namespace builds the reference model in the oracle harness.
    """
    a = dict(
        # data.py:552-577
        resolution=256, sequence_length=17, image_channels=3, sample_every_n_frames=1,
        # base.py:245-269
        embedding_dim=512, n_codes=8192, n_hiddens=240, lr=3e-4, downsample=(4, 4, 4),
        disc_channels=64, disc_layers=3, discriminator_iter_start=0, disc_loss_type="hinge",
        apply_allframes=False, image_gan_weight=1.0, video_gan_weight=1.0, l1_weight=4.0,
        gan_feat_weight=4.0, perceptual_weight=4.0, i3d_feat=False, restart_thres=1.0,
        no_random_restart=True, norm_type="batch", padding_type="replicate",
        # omnitokenizer.py:695-768
        lr_min=0.0, warmup_steps=0, warmup_lr_init=0.0, grad_accumulates=1, grad_clip_val=1.0,
        grad_clip_val_disc=1.0, disloss_check_thres=None, perloss_check_thres=None,
        recloss_check_thres=None, force_alternation=False, kl_weight=0.0, use_vae=False,
        video_perceptual_weight=0.0, initialize_vit=True, sigmoid_in_disc=False,
        activation_in_disc="leaky_relu", apply_blur=False, apply_noise=False, apply_diffaug=False,
        logitslaplace_weight=0.0, dis_warmup_steps=0, dis_lr_multiplier=1.0,
        dis_minlr_multiplier=False, recon_loss_type="l1", patch_size=8, gen_upscale=None,
        patch_embed="linear", enc_block="ttww", dec_block="tttt", twod_window_size=8,
        temporal_patch_size=4, defer_temporal_pool=False, defer_spatial_pool=False,
        spatial_pos="rope", spatial_depth=4, temporal_depth=4,
        causal_in_temporal_transformer=True, causal_in_peg=True, dim_head=64, heads=8,
        attn_dropout=0.0, ff_dropout=0.0, ff_mult=4.0, use_external_codebook=False,
        fp32_quant=False, codebook_type="vq", codebook_dim=8, l2_code=True,
        commitment_weight=1.0, resolution_scale=None,
    )
    if stage == 1:
        a.update(temporal_patch_size=2, spatial_pos="rel")
    elif stage != 2:
        raise ValueError("stage must be 1 or 2")
    a.update(overrides)
    return argparse.Namespace(**a)
