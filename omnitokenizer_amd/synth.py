"""Seeded synthetic weights and inputs for the encode/decode path.

No checkpoints or datasets are reachable offline (reference README.md:44-56 links only), so
parity tests, the bench and the golden fixtures use random weights of the released
architecture.  Everything here is generated with numpy's PCG64 (bit-stable across machines and
library versions) so that the build container (where the reference runs and the golden vectors
are made) and the GPU box (where the HIP path runs) see identical tensors.

`path_state_spec` enumerates the reference state_dict keys that belong to the path
(checked against the reference's own state_dict in
tests/test_oracle_vs_reference.py).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np
import torch

from .config import OmniTokConfig


def _transformer_spec(prefix: str, block: str, cfg: OmniTokConfig, spatial_pos: str, spec: OrderedDict):
    d, hd, heads = cfg.dim, cfg.dim_head, cfg.heads
    inner = cfg.ff_inner
    ws = cfg.window_size
    for i, c in enumerate(block):
        p = f"{prefix}.layers.{i}"
        if c == "t":
            spec[f"{p}.0.dsconv.weight"] = (d, 1, 3, 3, 3)
            spec[f"{p}.0.dsconv.bias"] = (d,)
            spec[f"{p}.1.q_scale"] = (hd,)
            spec[f"{p}.1.k_scale"] = (hd,)
            if spatial_pos == "rel":  # reference attention.py:363-364
                spec[f"{p}.1.spatial_rel_pos_bias.net.0.0.weight"] = (d, 2)
                spec[f"{p}.1.spatial_rel_pos_bias.net.0.0.bias"] = (d,)
                spec[f"{p}.1.spatial_rel_pos_bias.net.1.0.weight"] = (d, d)
                spec[f"{p}.1.spatial_rel_pos_bias.net.1.0.bias"] = (d,)
                spec[f"{p}.1.spatial_rel_pos_bias.net.2.weight"] = (heads, d)
                spec[f"{p}.1.spatial_rel_pos_bias.net.2.bias"] = (heads,)
            spec[f"{p}.1.norm.gamma"] = (d,)
            spec[f"{p}.1.norm.beta"] = (d,)
            spec[f"{p}.1.context_norm.gamma"] = (d,)
            spec[f"{p}.1.context_norm.beta"] = (d,)
            spec[f"{p}.1.to_q.weight"] = (hd * heads, d)
            spec[f"{p}.1.to_kv.weight"] = (2 * hd * heads, d)
            spec[f"{p}.1.to_out.weight"] = (d, hd * heads)
        elif c == "w":
            spec[f"{p}.1.relative_position_bias_table"] = ((2 * ws - 1) ** 2, heads)
            spec[f"{p}.1.relative_position_index"] = (ws * ws, ws * ws)
            spec[f"{p}.1.norm.gamma"] = (d,)
            spec[f"{p}.1.norm.beta"] = (d,)
            spec[f"{p}.1.qkv.weight"] = (3 * d, d)
            spec[f"{p}.1.proj.weight"] = (d, d)
            spec[f"{p}.1.proj.bias"] = (d,)
        elif c == "l":  # Pooling('l'): Linear(4*dim, dim), reference attention.py:92-93
            spec[f"{p}.1.pool.weight"] = (d, 4 * d)
            spec[f"{p}.1.pool.bias"] = (d,)
        elif c in "am":  # parameter-free AvgPool2d / MaxPool2d
            pass
        elif c == "n":  # Up('n'): nn.Upsample(2, 'nearest'), reference attention.py:119-120
            pass
        elif c == "r":  # Up('r'): Upsample -> Rearrange -> Linear(dim, dim), reference attention.py:122-127
            spec[f"{p}.1.up.2.weight"] = (d, d)
            spec[f"{p}.1.up.2.bias"] = (d,)
        else:
            raise NotImplementedError(c)
        spec[f"{p}.3.0.weight"] = (d,)
        spec[f"{p}.3.0.bias"] = (d,)
        spec[f"{p}.3.1.weight"] = (2 * inner, d)
        spec[f"{p}.3.4.weight"] = (d, inner)
    spec[f"{prefix}.norm_out.gamma"] = (d,)
    spec[f"{prefix}.norm_out.beta"] = (d,)


def path_state_spec(cfg: OmniTokConfig) -> "OrderedDict[str, tuple]":
    """name -> shape for every reference state_dict entry on the encode/decode path."""
    spec: OrderedDict = OrderedDict()
    c, d = cfg.image_channels, cfg.dim
    p, pt = cfg.enc_patch_size, cfg.enc_temporal_patch_size
    k0, k1 = c * p * p, c * p * p * pt
    cnn = cfg.patch_embed == "cnn"

    def batchnorm(prefix, n):  # SyncBatchNorm state (base.py:276)
        spec[f"{prefix}.weight"] = (n,)
        spec[f"{prefix}.bias"] = (n,)
        spec[f"{prefix}.running_mean"] = (n,)
        spec[f"{prefix}.running_var"] = (n,)
        spec[f"{prefix}.num_batches_tracked"] = ()

    for name, k, ptk in (("to_patch_emb_first_frame", k0, 1), ("to_patch_emb", k1, pt)):
        if cnn:  # reference omnitokenizer.py:823-836
            spec[f"encoder.{name}.0.weight"] = (d, c, ptk, p, p)
            spec[f"encoder.{name}.0.bias"] = (d,)
            batchnorm(f"encoder.{name}.1", d)
            continue
        spec[f"encoder.{name}.1.weight"] = (k,)
        spec[f"encoder.{name}.1.bias"] = (k,)
        spec[f"encoder.{name}.2.weight"] = (d, k)
        spec[f"encoder.{name}.2.bias"] = (d,)
        spec[f"encoder.{name}.3.weight"] = (d,)
        spec[f"encoder.{name}.3.bias"] = (d,)
    # temporal transformers are always built with the default spatial_pos="rel"
    # (reference omnitokenizer.py:860-861) and so carry unused spatial_rel_pos_bias weights
    _transformer_spec("encoder.enc_spatial_transformer", cfg.enc_block, cfg, cfg.spatial_pos, spec)
    _transformer_spec("encoder.enc_temporal_transformer", "t" * cfg.temporal_depth, cfg, "rel", spec)
    _transformer_spec("decoder.dec_spatial_transformer", cfg.dec_block, cfg, cfg.spatial_pos, spec)
    _transformer_spec("decoder.dec_temporal_transformer", "t" * cfg.temporal_depth, cfg, "rel", spec)
    pd, ptd = cfg.dec_patch_size, cfg.dec_temporal_patch_size
    for name, ptk in (("to_pixels_first_frame", 1), ("to_pixels", ptd)):
        if cnn:  # reference omnitokenizer.py:1019-1031
            spec[f"decoder.{name}.1.weight"] = (d, c, ptk, pd, pd)
            spec[f"decoder.{name}.1.bias"] = (c,)
            batchnorm(f"decoder.{name}.2", c)
        else:
            spec[f"decoder.{name}.0.weight"] = (c * pd * pd * ptk, d)
            spec[f"decoder.{name}.0.bias"] = (c * pd * pd * ptk,)
    if cfg.use_external_codebook:
        # VectorQuantize + CosineSimCodebook state (vector_quantize_pytorch.py:514-560, 690-860); pre_vq_conv /
        # post_vq_conv are nn.Identity (omnitokenizer.py:136-137): project_in / project_out take their place
        spec["codebook.codebook_usage"] = (cfg.n_codes,)
        spec["codebook.project_in.weight"] = (cfg.codebook_dim, d)
        spec["codebook.project_in.bias"] = (cfg.codebook_dim,)
        spec["codebook.project_out.weight"] = (d, cfg.codebook_dim)
        spec["codebook.project_out.bias"] = (d,)
        spec["codebook._codebook.initted"] = (1,)
        spec["codebook._codebook.cluster_size"] = (1, cfg.n_codes)
        spec["codebook._codebook.embed"] = (1, cfg.n_codes, cfg.codebook_dim)
        spec["codebook._codebook.embed_avg"] = (1, cfg.n_codes, cfg.codebook_dim)
        return spec
    spec["codebook.embeddings"] = (cfg.n_codes, cfg.codebook_dim)
    spec["codebook.N"] = (cfg.n_codes,)
    spec["codebook.z_avg"] = (cfg.n_codes, cfg.codebook_dim)
    spec["codebook.codebook_usage"] = (cfg.n_codes,)
    # --use_vae: pre_vq emits mean | logvar (reference omnitokenizer.py:149-153)
    pre_out = cfg.codebook_dim * (2 if cfg.use_vae else 1)
    spec["pre_vq_conv.1.weight"] = (pre_out, d)
    spec["pre_vq_conv.1.bias"] = (pre_out,)
    spec["post_vq_conv.1.weight"] = (d, cfg.codebook_dim)
    spec["post_vq_conv.1.bias"] = (d,)
    return spec


def relative_position_index(ws: int) -> torch.Tensor:
    """Swin-style index table, same arithmetic as reference attention.py:230-241:
    index[i,j] = (dy+ws-1)*(2ws-1) + (dx+ws-1), (dy,dx) = pos(i)-pos(j)."""
    yy, xx = np.divmod(np.arange(ws * ws), ws)
    dy = yy[:, None] - yy[None, :] + ws - 1
    dx = xx[:, None] - xx[None, :] + ws - 1
    return torch.from_numpy((dy * (2 * ws - 1) + dx).astype(np.int64))


def _key_seed(seed: int, name: str) -> int:
    return (seed * 1000003 + zlib.crc32(name.encode())) & 0xFFFFFFFF


def synth_state_dict(cfg: OmniTokConfig, seed: int = 0, profile: str = "default") -> "OrderedDict[str, torch.Tensor]":
    """Random but non-degenerate parameters: every gamma/bias/scale is perturbed so that a
    kernel that drops one of them is caught.

    profile "heavy": statistics a trained checkpoint can have and the default generator never produces -- heavy-tailed
    (Student-t, 3 degrees of freedom, unclipped) linear and depthwise-conv weights, LayerNorm gains with 1 % outlier
    channels of magnitude 5..20 (outlier activation channels in front of every GEMM), q/k scales up to 4 (logits up to
    8 * 4 * 4 = 128: near one-hot softmax rows), biases ten times larger.  This is the family that stresses the
    power-of-two operand scales of the fp16-split GEMMs and attention (gemm_h2.hip, gemm_pl.h, attn_h2.hip)."""
    if profile not in ("default", "heavy"):
        raise ValueError(f"unknown weight profile {profile!r}")
    heavy = profile == "heavy"
    sd: OrderedDict = OrderedDict()
    for name, shape in path_state_spec(cfg).items():
        rng = np.random.Generator(np.random.PCG64(_key_seed(seed, name + ("|heavy" if heavy else ""))))
        leaf = name.rsplit(".", 1)[-1]

        def randn(scale=1.0, mean=0.0):
            return (rng.standard_normal(shape, dtype=np.float32) * np.float32(scale)
                    + np.float32(mean)).astype(np.float32)

        def student_t(scale):
            return (rng.standard_t(3.0, size=shape) * scale).astype(np.float32)

        def gain():  # LayerNorm gains: 1 +- 0.1 with 1 % outlier channels of magnitude 5..20 and random sign
            v = randn(0.1, 1.0)
            n = v.size
            idx = rng.choice(n, size=max(1, n // 100), replace=False)
            v.reshape(-1)[idx] = (rng.uniform(5.0, 20.0, size=idx.size) * rng.choice([-1.0, 1.0], size=idx.size)).astype(np.float32)
            return v

        if name.endswith("relative_position_index"):
            sd[name] = relative_position_index(cfg.window_size)
            continue
        if leaf == "num_batches_tracked":
            sd[name] = torch.tensor(0, dtype=torch.int64)
            continue
        if leaf == "running_var":
            sd[name] = torch.from_numpy((rng.random(shape, dtype=np.float32) + np.float32(0.5)).astype(np.float32))
            continue
        if leaf == "running_mean":
            sd[name] = torch.from_numpy((rng.standard_normal(shape, dtype=np.float32) * np.float32(0.1)))
            continue
        if name == "codebook._codebook.embed":
            v = randn(1.0)
            if cfg.l2_code:  # CosineSimCodebook: l2norm(uniform_init(...)), vector_quantize_pytorch.py:539
                v = v / np.linalg.norm(v, axis=-1, keepdims=True)
        elif name == "codebook._codebook.embed_avg":
            v = sd["codebook._codebook.embed"].numpy().copy()
        elif name == "codebook._codebook.initted":
            v = np.ones(shape, np.float32)
        elif name == "codebook._codebook.cluster_size":
            v = np.zeros(shape, np.float32)
        elif name == "codebook.embeddings":
            v = randn(1.0)  # reference codebook.py:14 torch.randn(n_codes, dim)
        elif name == "codebook.z_avg":
            v = sd["codebook.embeddings"].numpy().copy()
        elif name in ("codebook.N", "codebook.codebook_usage"):
            v = np.zeros(shape, np.float32)
        elif heavy and "dsconv.weight" in name:
            v = student_t(0.5 / np.sqrt(27.0))
        elif heavy and "dsconv.bias" in name:
            v = randn(0.5)
        elif heavy and leaf in ("q_scale", "k_scale"):
            v = rng.uniform(0.5, 4.0, size=shape).astype(np.float32)
        elif heavy and leaf == "gamma":
            v = gain()
        elif heavy and leaf == "relative_position_bias_table":
            v = randn(2.0)
        elif heavy and "spatial_rel_pos_bias" not in name and leaf == "weight" and len(shape) == 1:
            v = gain()
        elif heavy and "spatial_rel_pos_bias" not in name and leaf == "bias":
            v = randn(0.5)
        elif heavy and "spatial_rel_pos_bias" not in name and leaf == "weight":
            v = student_t(0.02)
        elif "dsconv.weight" in name:
            v = (rng.random(shape, dtype=np.float32) * 2 - 1) * np.float32(1 / np.sqrt(27.0))
        elif "dsconv.bias" in name:
            v = randn(0.05)
        elif leaf in ("q_scale", "k_scale", "gamma"):
            v = randn(0.1, 1.0)
        elif leaf == "beta":
            v = np.zeros(shape, np.float32)  # fixed-zero buffer, reference attention.py:77
        elif leaf == "relative_position_bias_table":
            v = randn(0.5)
        elif "spatial_rel_pos_bias" in name:
            fan_in = shape[-1] if len(shape) > 1 else 1
            v = randn(1.0 / np.sqrt(fan_in)) if leaf == "weight" else randn(0.1)
        elif leaf == "weight" and len(shape) == 1:  # nn.LayerNorm weight
            v = randn(0.1, 1.0)
        elif leaf == "bias":
            v = randn(0.05)
        elif leaf == "weight":
            v = np.clip(randn(0.04), -0.08, 0.08)
        else:
            raise KeyError(name)
        sd[name] = torch.from_numpy(np.ascontiguousarray(v.astype(np.float32)))
    return sd


def synth_video(batch: int, frames: int, resolution: int, seed: int = 1234, smooth: bool = True,
                channels: int = 3, kind: str = "noise") -> torch.Tensor:
    """[B,C,F,H,W] fp32 in [-0.5, 0.5] (the reference's preprocess contract, data.py:346).
    smooth=True low-pass filters the noise so LayerNorm inputs are not white.
    kind "natural": image-like content -- strongly low-pass noise (large smooth regions, a few edges) with a slow
    drift over the frames; kind "mixed": clip 0 is ONE constant colour (zero-variance patches: LayerNorm's 1/sqrt(eps)
    branch), the others are "natural"."""
    if kind in ("natural", "mixed"):
        rng = np.random.Generator(np.random.PCG64(seed ^ 0x5EED))
        x = rng.standard_normal((batch, channels, 1, resolution, resolution)).astype(np.float32)
        for width in (resolution // 4, resolution // 8, 3):  # repeated box blurs: ~1/f^2 spectrum
            w = max(1, int(width))
            for ax in (3, 4):
                acc = np.zeros_like(x)
                for sh in range(-(w // 2), w - w // 2):
                    acc += np.roll(x, sh, ax)
                x = acc / np.float32(w)
        x = x + np.float32(0.15) * np.sign(x - np.median(x))  # a few hard edges
        drift = np.linspace(0.0, 1.0, frames, dtype=np.float32).reshape(1, 1, frames, 1, 1)
        x = x * (np.float32(1.0) - np.float32(0.3) * drift) + np.float32(0.05) * drift * np.roll(x, resolution // 8, 4)
        x = (x - x.min()) / (x.max() - x.min())
        if kind == "mixed":
            col = rng.random((channels, 1, 1, 1), dtype=np.float32)
            x[0] = np.broadcast_to(col, x[0].shape)
        return torch.from_numpy(np.ascontiguousarray((x - np.float32(0.5)).astype(np.float32)))
    if kind != "noise":
        raise ValueError(f"unknown input kind {kind!r}")
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.random((batch, channels, frames, resolution, resolution), dtype=np.float32)
    if smooth:
        for ax in (3, 4):
            x = (x + np.roll(x, 1, ax) + np.roll(x, -1, ax) + np.roll(x, 2, ax)) * np.float32(0.25)
        x = (x - x.min()) / (x.max() - x.min())
    return torch.from_numpy(np.ascontiguousarray((x - np.float32(0.5)).astype(np.float32)))


def synth_image(batch: int, resolution: int, seed: int = 1234, smooth: bool = True, kind: str = "noise") -> torch.Tensor:
    """[B,C,H,W]."""
    return synth_video(batch, 1, resolution, seed, smooth, kind=kind)[:, :, 0].contiguous()


def state_checksum(sd) -> int:
    """crc32 over all tensors in key order: pins the generator in the golden fixtures."""
    c = 0
    for k in sd:
        c = zlib.crc32(k.encode(), c)
        c = zlib.crc32(sd[k].detach().cpu().contiguous().numpy().tobytes(), c)
    return c


def synth_gpt_state(vocab_size, block_size, n_layer, n_head, n_embd, seed=0, vtokens_pos_shape=None):
    """Seeded numpy weights with the reference GPT's key names / shapes (gpt.py:172-193).
    vtokens_pos_shape = (sequence_length, resolution): adds vtokens_pos_emb [1, T, R, R, C] (gpt.py:183-184)."""
    sd = OrderedDict()

    def put(name, shape, kind):
        rng = np.random.Generator(np.random.PCG64((seed * 1000003 + zlib.crc32(name.encode())) & 0xFFFFFFFF))
        if kind == "w":
            v = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.05)
        elif kind == "b":
            v = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.02)
        elif kind == "g":
            v = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.1) + np.float32(1.0)
        else:
            v = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.5)
        sd[name] = torch.from_numpy(np.ascontiguousarray(v.astype(np.float32)))

    C = n_embd
    put("pos_emb", (1, block_size, C), "e")
    put("tok_emb.weight", (vocab_size, C), "e")
    for i in range(n_layer):
        p = f"blocks.{i}"
        put(f"{p}.ln1.weight", (C,), "g"); put(f"{p}.ln1.bias", (C,), "b")
        put(f"{p}.ln2.weight", (C,), "g"); put(f"{p}.ln2.bias", (C,), "b")
        for n in ("key", "query", "value", "proj"):
            put(f"{p}.attn.{n}.weight", (C, C), "w"); put(f"{p}.attn.{n}.bias", (C,), "b")
        put(f"{p}.mlp.0.weight", (4 * C, C), "w"); put(f"{p}.mlp.0.bias", (4 * C,), "b")
        put(f"{p}.mlp.2.weight", (C, 4 * C), "w"); put(f"{p}.mlp.2.bias", (C,), "b")
    put("ln_f.weight", (C,), "g"); put("ln_f.bias", (C,), "b")
    put("head.weight", (vocab_size, C), "w")
    if vtokens_pos_shape is not None:
        t, r = vtokens_pos_shape
        put("vtokens_pos_emb", (1, t, r, r, C), "e")
    return sd
