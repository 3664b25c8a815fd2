"""Drop-in for the reference's `OmniTokenizer_VQGAN` on the encode()/decode() path.

Mirrors reference OmniTokenizer/omnitokenizer.py: constructor from the argparse Namespace (:64),
`encode(x, is_image, include_embeddings=False)` (:247-266), `decode(encodings, is_image)`
(:268-317), the state_dict key names / shapes of the path and the attributes the
reference's callers touch (vqgan_eval.py:76-86, lm_transformer.py:95-101).  All arithmetic runs in
libomnitok.so (hand-written gfx950 HIP kernels behind the C ABI of include/omnitok.h); this class
only owns the parameters (as torch tensors, so .to()/.state_dict()/.load_state_dict() behave) and
hands device pointers to the native engine.  Inference only; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
import itertools
import math
import weakref
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import OmnitokConfig, check
from .config import OmniTokConfig
from .synth import path_state_spec, relative_position_index

_OFF_PATH_PREFIXES = ("image_discriminator.", "video_discriminator.", "perceptual_model.")


class _Holder(nn.Module):
    """Parameter container that reproduces the reference's module nesting (names only)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container; call OmniTokenizer_VQGAN.encode()/decode()")


def _attach(root: nn.Module, dotted: str, tensor: torch.Tensor):
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Holder())
        mod = mod._modules[p]
    if tensor.dtype.is_floating_point:
        mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))
    else:
        mod.register_buffer(parts[-1], tensor)


def _on_own_device(fn):
    """Runs a method with the module's GPU as the current device: the native calls allocate and launch on
    the CURRENT device / its current stream, so a model on cuda:1 must not run while cuda:0 is current."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError(
                f"OmniTokenizer_VQGAN is on {dev}: move it to the GPU (.to('cuda')). The MI355X HIP path is "
                "the only implementation; there is no CPU fallback.")
        with torch.cuda.device(dev):
            return fn(self, *a, **k)
    return wrapper


# modules by handle: the omnitok::vqgan_encode / vqgan_decode custom ops take tensors and an integer handle
_MODELS: "weakref.WeakValueDictionary[int, OmniTokenizer_VQGAN]" = weakref.WeakValueDictionary()
_HANDLES = itertools.count(1)


def _model(handle: int) -> "OmniTokenizer_VQGAN":
    m = _MODELS.get(handle)
    if m is None:
        raise RuntimeError(f"omnitok::vqgan_*: no live OmniTokenizer_VQGAN with handle {handle}")
    return m


def load_checkpoint_file(path, map_location="cpu", trust=False):
    """torch.load of a PL-format checkpoint WITHOUT executing pickled code: weights_only=True with argparse.Namespace (the
    `hyper_parameters.args` object of the released checkpoints, reference omnitokenizer.py:208) allow-listed.  A file that needs
    anything else is refused unless the caller vouches for it (trust=True / --trust-checkpoint / OMNITOK_TRUST_CHECKPOINT=1): the
    reference's own loader (pytorch_lightning) unpickles arbitrary objects, a downloaded .ckpt should not get that by default."""
    import argparse
    import os
    import pickle
    try:
        with torch.serialization.safe_globals([argparse.Namespace]):
            return torch.load(path, map_location=map_location, weights_only=True)
    except pickle.UnpicklingError as e:
        if not (trust or os.environ.get("OMNITOK_TRUST_CHECKPOINT") == "1"):
            raise RuntimeError(f"{path}: not loadable with weights_only=True ({e}); pass trust_checkpoint=True (tools/ckpt_parity.py "
                               "--trust-checkpoint) only for a file whose origin you trust -- unpickling runs its code") from e
        return torch.load(path, map_location=map_location, weights_only=False)


class OmniTokenizer_VQGAN(nn.Module):
    def __init__(self, args, attention_mode: Optional[str] = None):
        """args: the same Namespace the reference takes.  attention_mode: "sdpa" (what the
        reference executes under torch >= 2.1, attention.py:439) or "legacy" (the einsum branch
        `imagenet_only.ckpt` was trained with, README.md:58); default "sdpa", or
        args.attention_mode if present."""
        super().__init__()
        mode = attention_mode or getattr(args, "attention_mode", "sdpa")
        self.args = args
        self.cfg = OmniTokConfig.from_args(args, attention_mode=mode)
        cfg = self.cfg
        # attributes read by the reference's callers
        self.embedding_dim = cfg.dim
        self.n_codes = cfg.n_codes
        self.resolution = cfg.resolution
        self.patch_size = cfg.patch_size
        self.use_vae = cfg.use_vae
        self.kl_weight = getattr(args, "kl_weight", 0.0)
        self.use_external_codebook = cfg.use_external_codebook
        self.l2_code = cfg.l2_code

        for name, shape in path_state_spec(cfg).items():
            if name.endswith("relative_position_index"):
                t = relative_position_index(cfg.window_size)
            elif name.endswith("num_batches_tracked"):
                t = torch.tensor(0, dtype=torch.int64)
            elif name.endswith((".gamma", ".q_scale", ".k_scale", ".running_var")) or \
                    (name.endswith(".weight") and len(shape) == 1):
                t = torch.ones(shape)
            else:
                t = torch.zeros(shape)
            _attach(self, name, t)
        self.encoder.image_size = (cfg.resolution, cfg.resolution)
        self.decoder.image_size = (cfg.resolution, cfg.resolution)
        self.codebook.n_codes = cfg.n_codes
        self.codebook.codebook_size = cfg.n_codes  # VectorQuantize's name for it
        self.codebook.embedding_dim = cfg.codebook_dim
        self.codebook._need_init = False  # training-time k-means init (codebook.py:40-51) is not on the path

        self.codebook.call_cnt = 0          # reference codebook.py:19 (usage EMA state)
        self.codebook.usage_sigma = 0.99    # reference codebook.py:12
        # Reference Codebook.forward updates the `codebook_usage` EMA buffer and `call_cnt` on EVERY call, also in eval mode
        # and therefore on every encode() (codebook.py:122-143): a state_dict() taken after N encodes carries that state.
        # True (default) reproduces it with one histogram kernel per encode (microseconds); False makes encode() free of
        # side effects (forward(log_image=True) still updates, as the statistics are part of what it returns).
        self.update_codebook_usage_on_encode = True

        self._engine = None
        self._engine_sig = None
        self._engine_dev = None
        self._weights_epoch = getattr(self, "_weights_epoch", 0)
        self._timing = False
        self._workspace = None  # uint8 tensor of PyTorch's caching allocator lent to the engine (grow-only)
        self._workspace_stream = None
        self._handle = next(_HANDLES)  # how the omnitok::vqgan_* custom ops find this module
        _MODELS[self._handle] = self

    # ---- nn.Module plumbing -------------------------------------------------------------------
    @property
    def device(self):
        return self.encoder.enc_spatial_transformer.norm_out.gamma.device

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("omnitokenizer_amd is inference-only (encode/decode path)")
        return super().train(False)

    def load_state_dict(self, state_dict, strict: Optional[bool] = None, assign: bool = False):
        """Accepts a reference checkpoint's state_dict.  Off-path entries (discriminators, LPIPS;
        vqgan_eval.py:62-68 drops or ignores them the same way) are skipped silently.  A tensor the
        encode/decode path reads that is MISSING from the state_dict raises (every parameter is pre-created
        here, so a silently missing key -- e.g. a 'module.' prefix -- would encode with placeholder weights);
        pass strict=False explicitly to get nn.Module's lenient behaviour (a warning is issued)."""
        sd = {k: v for k, v in state_dict.items() if not k.startswith(_OFF_PATH_PREFIXES)}
        out = super().load_state_dict(sd, strict=bool(strict), assign=assign)
        self._engine_sig = None
        self._weights_epoch += 1
        if out.missing_keys:
            msg = (f"state_dict lacks {len(out.missing_keys)} tensors of the encode/decode path, e.g. "
                   f"{out.missing_keys[:4]}")
            if strict is None:
                raise RuntimeError(msg + " (pass strict=False to load anyway)")
            import warnings
            warnings.warn(msg)
        return out

    @classmethod
    def load_from_checkpoint(cls, path, map_location="cpu", strict=False, **kw):
        """PL-style checkpoint: {"state_dict":…, "hyper_parameters": {"args": Namespace}}
        (reference omnitokenizer.py:208, download.py:49)."""
        ckpt = load_checkpoint_file(path, map_location=map_location, trust=kw.pop("trust_checkpoint", False))
        model = cls(ckpt["hyper_parameters"]["args"], **kw)
        model.load_state_dict(ckpt["state_dict"], strict=strict)
        return model

    @property
    def latent_shape(self):
        """reference omnitokenizer.py:239-245: (frames, H, W) // args.downsample."""
        a = self.args
        inp = (a.sequence_length // a.sample_every_n_frames, a.resolution, a.resolution)
        return tuple(s // d for s, d in zip(inp, a.downsample))

    def __del__(self):
        try:
            if self._engine is not None:
                _lib.load().omnitok_engine_destroy(self._engine)
        except Exception:
            pass

    # ---- native engine ------------------------------------------------------------------------
    def _native_config(self) -> OmnitokConfig:
        c = self.cfg
        nc = OmnitokConfig()
        nc.resolution, nc.image_channels, nc.patch_size = c.resolution, c.image_channels, c.patch_size
        nc.temporal_patch_size, nc.dim, nc.heads, nc.dim_head = c.temporal_patch_size, c.dim, c.heads, c.dim_head
        nc.ff_inner, nc.window_size, nc.n_codes, nc.codebook_dim = c.ff_inner, c.window_size, c.n_codes, c.codebook_dim
        nc.l2_code = int(c.l2_code)
        nc.spatial_rope = int(c.spatial_pos == "rope")
        nc.legacy_attention = int(c.attention_mode == "legacy")
        nc.causal_temporal = int(c.causal_in_temporal_transformer)
        nc.causal_peg = int(c.causal_in_peg)
        nc.temporal_depth = c.temporal_depth
        nc.enc_block = c.enc_block.encode()
        nc.dec_block = c.dec_block.encode()
        nc.use_vae = int(c.use_vae)
        nc.patch_embed_cnn = int(c.patch_embed == "cnn")
        nc.defer_temporal_pool = int(c.defer_temporal_pool)
        nc.defer_spatial_pool = int(c.defer_spatial_pool)
        nc.gen_upscale = int(c.gen_upscale)
        nc.external_codebook = int(c.use_external_codebook)
        return nc

    def _signature(self):
        """Change detector for the engine's weight copies: an epoch counter bumped by load_state_dict / _apply (.to(),
        .cuda(), .float()) plus the storage pointer and in-place version of EVERY parameter and buffer on the path
        -- in-place edits THROUGH THE PARAMETER under no_grad (p.mul_(), p.copy_(), torch.nn.init.*) and re-assigned
        storage are seen.  NOT seen: edits through `p.data` (p.data.copy_(), p.data.mul_(), the usual EMA store / restore
        idiom) -- `.data` is a fresh tensor object with its own version counter, so neither the pointer nor p._version
        moves; after such an edit call mark_weights_changed(), or the engine keeps computing with its old copies
        (tests/test_gpu_e2e.py::test_data_edits_need_mark_weights_changed pins both halves).  A device-side checksum
        would catch them at the price of a kernel and a host synchronisation per call, which the path does not pay.
        The tensor list is cached per epoch: reading ~300 (data_ptr, _version) pairs costs tens of microseconds."""
        cached = getattr(self, "_sig_tensors", None)
        if cached is None or cached[0] != self._weights_epoch:
            cached = (self._weights_epoch, [t for t in list(self.parameters()) + list(self.buffers())])
            self._sig_tensors = cached
        return (self._weights_epoch,) + tuple((t.data_ptr(), t._version) for t in cached[1])

    def set_option(self, name: str, value: int):
        """Arithmetic / data-flow mode of THIS module's engine ("gemm_mode", "attn_mode", "attn_vpack", "gemm_pl"; -1 =
        follow the process default of _lib.set_option again): modules of one process can run different modes."""
        self._sync_engine()
        check(_lib.load().omnitok_engine_set_option(self._engine, name.encode(), int(value)), "engine_set_option")

    def mark_weights_changed(self):
        """Call after editing parameters through `.data` (p.data.copy_(), EMA swaps, ...) so that the next encode / decode
        re-uploads them; edits through the parameter itself, load_state_dict and .to() are detected without it."""
        self._weights_epoch += 1

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._weights_epoch = getattr(self, "_weights_epoch", 0) + 1
        return out

    def _sync_engine(self):
        """Pushes the parameters into the native engine when they changed (load_state_dict, .to(),
        in-place edits)."""
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError(
                f"OmniTokenizer_VQGAN is on {dev}: move it to the GPU (.to('cuda')). The MI355X HIP path is "
                "the only implementation; there is no CPU fallback.")
        if self._engine is not None and self._engine_dev != dev:
            raise RuntimeError(f"the native engine was built on {self._engine_dev}; the module now lives on {dev} "
                               "(build a new OmniTokenizer_VQGAN for another GPU)")
        sig = self._signature()
        if self._engine is not None and sig == self._engine_sig:
            return
        lib = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream
        if self._engine is None:
            h = ctypes.c_void_p()
            nc = self._native_config()
            check(lib.omnitok_engine_create(ctypes.byref(nc), ctypes.byref(h)), "engine_create")
            self._engine = h
            self._engine_dev = dev
            lib.omnitok_engine_set_timing(self._engine, int(self._timing))
        for name, t in self.state_dict(keep_vars=True).items():
            t = t.detach()
            if t.dtype.is_floating_point and t.dtype != torch.float32:
                raise TypeError(f"{name}: parameters must be float32 (the path computes in fp32 like the reference)")
            t = t.contiguous()
            shape = (ctypes.c_int64 * max(t.dim(), 1))(*t.shape)
            check(lib.omnitok_engine_set_weight(self._engine, name.encode(), ctypes.c_void_p(t.data_ptr()), shape,
                                                t.dim(), int(t.dtype == torch.int64), stream), f"set_weight({name})")
        check(lib.omnitok_engine_finalize(self._engine, stream), "engine_finalize")
        torch.cuda.current_stream().synchronize()
        self._engine_sig = sig

    def _lend_workspace(self, need):
        """The engine's activations live in a tensor of PyTorch's caching allocator (visible to
        torch.cuda.memory_allocated, released with the module), grown when a larger call arrives."""
        if need < 0:
            return  # invalid shape: the native call reports it
        cur = torch.cuda.current_stream()
        if self._workspace is not None and self._workspace.numel() >= need:
            # used from a stream other than the one it was allocated on: tell the caching allocator, so that it
            # does not recycle the block under a running kernel once the module is gone
            if torch.cuda.is_current_stream_capturing():
                self._workspace_captured = True  # a HIP graph now holds this block's addresses
            elif cur != self._workspace_stream:
                self._workspace.record_stream(cur)
            return
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the engine workspace has to grow for this shape, which needs a synchronisation: run "
                               "one eager encode/decode of the same shape before capturing a HIP graph")
        cur.synchronize()  # kernels of earlier calls may still read the old block
        if getattr(self, "_workspace_captured", False):
            # a captured graph replays with the old block's addresses baked in: keep it alive for the module's lifetime
            # instead of returning it to the caching allocator (a later replay would read and write recycled memory)
            self._retired_workspaces = getattr(self, "_retired_workspaces", []) + [self._workspace]
            self._workspace_captured = False
        self._workspace = None
        self._workspace = torch.empty(int(need * 1.02) + 256, dtype=torch.uint8, device=self.device)
        self._workspace_stream = cur
        ptr = (self._workspace.data_ptr() + 255) // 256 * 256
        check(_lib.load().omnitok_engine_set_workspace(self._engine, ctypes.c_void_p(ptr),
                                                       self._workspace.numel() - (ptr - self._workspace.data_ptr())),
              "set_workspace")

    def latent_dims(self, F, H, W):
        """(T', h, w) of encode() for [.., F, H, W] pixels from the configuration alone (no engine): what the
        omnitok::vqgan_encode fake implementation and shape checks use."""
        c = self.cfg
        t = 1 + (F - 1) // c.enc_temporal_patch_size
        h, w = H // c.enc_patch_size, W // c.enc_patch_size
        for ch in c.enc_block:
            if ch in "aml":
                h, w = h // 2, w // 2
            elif ch in "nr":
                h, w = h * 2, w * 2
        if c._defer_s:
            h, w = h // 2, w // 2
        if c._defer_t:
            t = 1 + (t - 1) // 2
        return t, h, w

    def pixel_dims(self, T, h, w):
        """(F, H, W) of decode() for [.., T', h, w] latents from the configuration alone."""
        c = self.cfg
        t = 1 + (T - 1) * 2 if c._defer_t else T
        up = 2 if c._defer_s else 1
        return 1 + (t - 1) * c.dec_temporal_patch_size, h * up * c.dec_patch_size, w * up * c.dec_patch_size

    # ---- the path -------------------------------------------------------------------------------
    @torch.no_grad()
    @_on_own_device
    def encode(self, x, is_image, include_embeddings=False, return_latents=False, noise=None,
               sample_posterior=True, return_moments=False, _stats_out=None):
        """reference omnitokenizer.py:247-266.  x: [B,C,H,W] (is_image) or [B,C,F,H,W] fp32 in
        [-0.5,0.5] on the GPU.  Returns LongTensor ids [B,T',h,w]; with include_embeddings
        (embeddings [B,cdim,T',h,w], ids).

        With args.use_vae (:260-266) returns the posterior sample z [B,cdim,T',h,w]
        ([B,cdim,h,w] for images).  The noise is drawn like the reference does
        (modules/vae.py:16: torch.randn on the host with the global generator, then moved), so
        torch.manual_seed() reproduces the reference's sample; pass `noise` ([B,cdim,T',h,w],
        any device) to supply it, or sample_posterior=False for posterior.mode().
        return_moments adds the raw mean|logvar tensor [B,2*cdim,T',h,w]."""
        if x.dim() not in (4, 5):
            raise AssertionError("video.ndim in {4, 5}")  # reference omnitokenizer.py:921
        if is_image:
            if x.dim() != 4:
                raise ValueError("is_image=True expects [B,C,H,W]")
            B, C, H, W = x.shape
            F = 1
        else:
            if x.dim() != 5:
                raise ValueError("is_image=False expects [B,C,F,H,W]")
            B, C, F, H, W = x.shape
        if C != self.cfg.image_channels:
            raise ValueError(f"expected {self.cfg.image_channels} channels, got {C}")
        self._sync_engine()
        if x.device != self.device:
            raise RuntimeError(f"input on {x.device}, model on {self.device}")
        x = x.to(torch.float32).contiguous()
        pt = self.cfg.enc_temporal_patch_size
        if (F - 1) % pt != 0:
            raise AssertionError(f"number of frames ({F}) minus one ({F - 1}) must be divisible by temporal "
                                 f"patch size ({pt})")  # reference omnitokenizer.py:931-932
        T, h, w = self._shape("encode", F, H, W)
        if self.use_vae:
            self._lend_workspace(_lib.load().omnitok_engine_workspace_need_encode(self._engine, B, F, H, W))
            return self._encode_vae(x, is_image, (B, F, H, W, T, h, w), noise, sample_posterior, return_moments)
        # through the registered operator (omnitok::vqgan_encode): torch.compile / export see an op with a
        # shape function instead of opaque Python
        x5 = x if x.dim() == 5 else x[:, :, None]
        ids, emb, z = torch.ops.omnitok.vqgan_encode(x5, self._handle, bool(include_embeddings), bool(return_latents))
        # decode() trusts THIS tensor object (no range check read-back) for as long as it is alive and unmodified.  By
        # identity, not by address: the caching allocator hands a freed block to the next tensor of the same size, and
        # every fresh tensor has _version 0, so (data_ptr, _version, shape) would also match foreign ids.
        self._own_ids = (weakref.ref(ids), ids._version)
        # the eval-time state mutation of reference Codebook.forward (codebook.py:122-143); forward(log_image=True) passes
        # _stats_out to receive the statistics of this one update.  Not recorded into a HIP graph: `call_cnt` is host state, so
        # a captured encode() leaves codebook_usage / call_cnt untouched, at capture and at every replay (forward() refuses capture).
        if not self.use_external_codebook and (_stats_out is not None or self.update_codebook_usage_on_encode) \
                and ids.numel() > 0 and not torch.cuda.is_current_stream_capturing():
            stats = self._update_codebook_usage(ids)
            if _stats_out is not None:
                _stats_out.update(stats)
        if not include_embeddings:
            emb = None
        elif self.use_external_codebook:
            emb = emb.permute(0, 4, 1, 2, 3)  # 'b (t h w) c -> b c t h w', vector_quantize_pytorch.py:1077
        if return_latents:
            return (emb, ids, z) if include_embeddings else (ids, z)
        return (emb, ids) if include_embeddings else ids

    def _update_codebook_usage(self, ids):
        """batch_usage / perplexity / avg_usage of reference Codebook.forward (codebook.py:122-143) for `ids`, with its side
        effects: `codebook_usage` <- usage (first call) or sigma * codebook_usage + (1 - sigma) * usage; call_cnt += 1."""
        from . import ops
        cb = self.codebook
        usage, perplexity, avg_usage = ops.vq_stats(ids, self.cfg.n_codes, cb.codebook_usage.data, cb.call_cnt == 0,
                                                    cb.usage_sigma)
        cb.call_cnt += 1
        return dict(batch_usage=usage, perplexity=perplexity, avg_usage=avg_usage)

    def _encode_native(self, x, want_emb, want_z):
        """omnitok::vqgan_encode on CUDA tensors: x [B, C, F, H, W] fp32 contiguous -> (ids, emb, z); emb / z are
        empty tensors when not requested."""
        self._sync_engine()
        B, C, F, H, W = x.shape
        T, h, w = self._shape("encode", F, H, W)
        lib = _lib.load()
        self._lend_workspace(lib.omnitok_engine_workspace_need_encode(self._engine, B, F, H, W))
        ids = torch.empty(B, T, h, w, device=x.device, dtype=torch.int64)
        if not want_emb:
            emb = x.new_empty(0)
        elif self.use_external_codebook:  # project_out(embed[ids]), token-major from the engine
            emb = torch.empty(B, T, h, w, self.cfg.dim, device=x.device)
        else:
            emb = torch.empty(B, self.cfg.codebook_dim, T, h, w, device=x.device)
        z = torch.empty(B, T, h, w, self.cfg.codebook_dim, device=x.device) if want_z else x.new_empty(0)
        check(lib.omnitok_encode(self._engine, ctypes.c_void_p(x.data_ptr()), B, F, H, W,
                                 ctypes.c_void_p(ids.data_ptr()),
                                 ctypes.c_void_p(emb.data_ptr()) if want_emb else None,
                                 ctypes.c_void_p(z.data_ptr()) if want_z else None,
                                 torch.cuda.current_stream().cuda_stream), "encode")
        return ids, emb, z

    def _shape(self, which, a, b, c):
        """latent <-> pixel shapes from the engine (pooling blocks, deferred pools and gen_upscale
        change them; include/omnitok.h omnitok_engine_{encode,decode}_shape)."""
        out = [ctypes.c_int() for _ in range(3)]
        fn = getattr(_lib.load(), f"omnitok_engine_{which}_shape")
        check(fn(self._engine, a, b, c, *[ctypes.byref(o) for o in out]), f"{which}_shape")
        return tuple(o.value for o in out)

    def _encode_vae(self, x, is_image, dims, noise, sample_posterior, return_moments):
        B, F, H, W, T, h, w = dims
        cd = self.cfg.codebook_dim
        if sample_posterior:
            if noise is None:
                noise = torch.randn(B, cd, T, h, w)  # host draw, reference modules/vae.py:16
            if tuple(noise.shape) != (B, cd, T, h, w) and not (is_image and tuple(noise.shape) == (B, cd, h, w)):
                raise ValueError(f"noise must be [B,{cd},T',h,w] = {(B, cd, T, h, w)}, got {tuple(noise.shape)}")
            noise = noise.to(device=x.device, dtype=torch.float32).contiguous()
        else:
            noise = None
        z = torch.empty(B, cd, T, h, w, device=x.device, dtype=torch.float32)
        mom = torch.empty(B, 2 * cd, T, h, w, device=x.device, dtype=torch.float32) if return_moments else None
        check(_lib.load().omnitok_encode_vae(self._engine, ctypes.c_void_p(x.data_ptr()), B, F, H, W,
                                             None if noise is None else ctypes.c_void_p(noise.data_ptr()),
                                             ctypes.c_void_p(z.data_ptr()),
                                             None if mom is None else ctypes.c_void_p(mom.data_ptr()),
                                             torch.cuda.current_stream().cuda_stream), "encode_vae")
        if is_image:
            z = z.squeeze(2)  # b c t h w -> b c h w, reference omnitokenizer.py:264
        return (z, mom) if return_moments else z

    def _decode_vae(self, z, is_image):
        """reference omnitokenizer.py:293-317: image latents [B,hw,c] or channel-first [B,c,h,w];
        video latents [B,thw,c] or channel-LAST [B,t,h,w,c] (what Latte's sampler hands over,
        sample_ddp.py:201-203) -- encode()'s [B,c,t,h,w] must be permuted by the caller exactly
        as with the reference."""
        if z.device != self.device:
            raise RuntimeError(f"latents on {z.device}, model on {self.device}")
        z = z.to(torch.float32)
        cd = self.cfg.codebook_dim
        if z.dim() == 3:
            if is_image:
                h = int(math.sqrt(z.shape[1]))
                z = z.reshape(z.shape[0], 1, h, -1, z.shape[-1])
            else:
                h = self.resolution // self.patch_size
                z = z.reshape(z.shape[0], -1, h, h, z.shape[-1])
            channel_first = 0
        elif is_image:
            if z.dim() != 4:
                raise ValueError("is_image=True expects latents [B,c,h,w] or [B,hw,c]")
            z = z.unsqueeze(2)
            channel_first = 1
        else:
            if z.dim() != 5:
                raise ValueError("is_image=False expects latents [B,t,h,w,c] or [B,thw,c]")
            channel_first = 0
        z = z.contiguous()
        if channel_first:
            B, C, T, h, w = z.shape
        else:
            B, T, h, w, C = z.shape
        if C != cd:
            raise ValueError(f"latent channel dim is {C}, expected codebook_dim={cd} "
                             f"({'channel-first' if channel_first else 'channel-last'} layout)")
        c = self.cfg
        F, Ho, Wo = self._shape("decode", T, h, w)
        self._lend_workspace(_lib.load().omnitok_engine_workspace_need_decode(self._engine, B, T, h, w))
        out = torch.empty(B, c.image_channels, F, Ho, Wo, device=z.device, dtype=torch.float32)
        check(_lib.load().omnitok_decode_vae(self._engine, ctypes.c_void_p(z.data_ptr()), channel_first, B, T, h, w,
                                             ctypes.c_void_p(out.data_ptr()),
                                             torch.cuda.current_stream().cuda_stream), "decode_vae")
        return out[:, :, 0] if is_image else out

    @torch.no_grad()
    @_on_own_device
    def decode(self, encodings, is_image, check_ids: Optional[bool] = None):
        """reference omnitokenizer.py:268-291 (with --use_external_codebook the reference's decode() raises
        -- it reads codebook.embeddings, which VectorQuantize lacks -- so this computes what its forward()
        computes from the same ids: decoder(project_out(embed[ids]))).  encodings: ids [B,T',h,w], flat video ids
        [B,T'*h*w] (h = w = args.resolution // patch_size, :283-286) or flat image ids [B,h*w]
        (h = int(sqrt(h*w)), :272-275).  Returns [B,3,H,W] (is_image) or [B,3,F,H,W]."""
        self._sync_engine()
        if self.use_vae:
            return self._decode_vae(encodings, is_image)
        ids = encodings
        if ids.dtype != torch.int64:
            ids = ids.long()
        if ids.device != self.device:
            raise RuntimeError(f"ids on {ids.device}, model on {self.device}")
        if ids.dim() == 2:
            if is_image:
                h = int(math.sqrt(ids.shape[1]))
                ids = ids.reshape(ids.shape[0], 1, h, -1)
            else:
                h = self.resolution // self.patch_size
                ids = ids.reshape(ids.shape[0], -1, h, h)
        elif ids.dim() != 4:
            raise ValueError("encodings must be [B,T,h,w] or flat [B,N]")
        ids = ids.contiguous()
        B, T, h, w = ids.shape
        if is_image and T != 1:
            raise ValueError("is_image=True expects a single latent frame")
        out = torch.ops.omnitok.vqgan_decode(ids, self._handle)
        lib = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream
        if check_ids is None:
            # like the reference's F.embedding, out-of-range ids raise; the check is one int read-back (a
            # host synchronisation), so it is skipped by default while a HIP graph is being captured and for the
            # untouched output of this module's own encode() (its ids are in range by construction)
            own = getattr(self, "_own_ids", None)
            trusted = own is not None and own[0]() is encodings and own[1] == encodings._version
            check_ids = not torch.cuda.is_current_stream_capturing() and not trusted
        if check_ids:
            rc = lib.omnitok_engine_check_ids(self._engine, stream)
            if rc == -1:
                raise IndexError(lib.omnitok_last_error().decode())
            check(rc, "check_ids")
        return out[:, :, 0] if is_image else out

    def _decode_native(self, ids):
        """omnitok::vqgan_decode on CUDA tensors: ids [B, T', h, w] int64 contiguous -> pixels [B, C, F, H, W]."""
        self._sync_engine()
        B, T, h, w = ids.shape
        F, Ho, Wo = self._shape("decode", T, h, w)
        lib = _lib.load()
        self._lend_workspace(lib.omnitok_engine_workspace_need_decode(self._engine, B, T, h, w))
        out = torch.empty(B, self.cfg.image_channels, F, Ho, Wo, device=ids.device, dtype=torch.float32)
        check(lib.omnitok_decode(self._engine, ctypes.c_void_p(ids.data_ptr()), B, T, h, w,
                                 ctypes.c_void_p(out.data_ptr()), torch.cuda.current_stream().cuda_stream), "decode")
        return out

    def forward(self, x, optimizer_idx=None, log_image=False):
        """The inference use of reference VQGAN.forward (omnitokenizer.py:330-413,
        `vqgan(x, log_image=True)` in vqgan_eval.py:119,187): returns
        (frames, frames_recon, x, x_recon, vq_output) with vq_output carrying encodings/embeddings."""
        if not log_image:
            raise NotImplementedError("training forward (losses, discriminators) is outside the built path")
        is_image = x.dim() == 4
        if self.use_vae:
            # reference omnitokenizer.py:366-370, 410-411: sample -> post_vq -> decoder, vq_output None
            z = self.encode(x, is_image)
            x_recon = self.decode(z if is_image else z.permute(0, 2, 3, 4, 1), is_image)
        else:
            stats = {}
            if torch.cuda.is_current_stream_capturing():
                # the statistics are part of this call's return value and advance host-side state (call_cnt): a captured graph
                # would return nothing for them and its replays would not advance it.  encode() / decode() alone
                # can be captured -- a replay then leaves codebook_usage where the capture found it.
                raise RuntimeError("forward(log_image=True) returns codebook statistics and updates call_cnt on the host: it cannot "
                                   "be recorded into a HIP graph (capture encode() / decode() instead)")
            emb, ids = self.encode(x, is_image, include_embeddings=True, _stats_out=None if self.use_external_codebook else stats)
            x_recon = self.decode(ids, is_image)
            if not self.use_external_codebook and not stats:   # empty batch: no update happened, the statistics are those of no ids
                z0 = torch.zeros(self.cfg.n_codes, device=x.device)
                stats = dict(batch_usage=z0, perplexity=torch.ones((), device=x.device), avg_usage=self.codebook.codebook_usage.data.clone())
        if is_image:
            frames, frames_recon = x, x_recon
        else:
            B, C, T, H, W = x.shape
            idx = torch.randint(0, T, [B], device=x.device).reshape(-1, 1, 1, 1, 1).repeat(1, C, 1, H, W)
            frames = torch.gather(x, 2, idx).squeeze(2)
            frames_recon = torch.gather(x_recon, 2, idx).squeeze(2)
        if self.use_vae:
            return frames, frames_recon, x, x_recon, None
        # the statistics Codebook.forward returns next to the ids (reference codebook.py:122-143);
        # vqgan_eval.py:152,195 accumulates batch_usage
        if self.use_external_codebook:  # VectorQuantize keeps no usage state; it reports its (eval: zero) commitment loss
            stats = self._update_codebook_usage(ids)
            vq_output = dict(embeddings=emb, encodings=ids, **stats)
            vq_output["commitment_loss"] = torch.zeros(1, device=x.device)
            vq_output["perplexity"] = self._ext_perplexity(ids)
        else:
            vq_output = dict(embeddings=emb, encodings=ids, **stats)
        return frames, frames_recon, x, x_recon, vq_output

    @staticmethod
    def _ext_perplexity(ids):
        """VectorQuantize.get_perplexity (vector_quantize_pytorch.py:849-853) one-hots the UNFLATTENED
        [b,t,h,w] indices and averages over dim 0 only, so its "perplexity" is exp of the summed per-position
        entropies of the batch histogram (2^64 for two different 8x8 images), not the codebook perplexity.
        Reproduced as such (a statistic of forward(), off the encode/decode path): with c[b,p] = how many
        batch items share item b's code at position p, -sum p log(p + 1e-10) = -(1/B) sum_{b,p} log(c/B + 1e-10)."""
        B = ids.shape[0]
        flat = ids.reshape(B, -1)
        total = torch.zeros((), device=ids.device, dtype=torch.float32)
        for b in range(B):  # O(B^2 P) compares, no [B, P, n_codes] one-hot
            cnt = (flat == flat[b:b + 1]).sum(0).to(torch.float32)
            total = total + torch.log(cnt / B + 1e-10).sum()
        return torch.exp(-total / B)

    # ---- measurement hooks ----------------------------------------------------------------------
    def set_timing(self, enabled: bool):
        self._timing = bool(enabled)
        if self._engine is not None:
            _lib.load().omnitok_engine_set_timing(self._engine, int(enabled))

    def timing_report(self) -> Dict[str, dict]:
        """{kernel family: {"calls", "ms", "work"}} accumulated since the last report (HIP events
        recorded around each launch on the launch stream; synchronises)."""
        if self._engine is None:
            return {}
        buf = ctypes.create_string_buffer(1 << 16)
        check(_lib.load().omnitok_engine_timing_report(self._engine, buf, len(buf)), "timing_report")
        out = {}
        for line in buf.value.decode().splitlines():
            name, calls, ms, work = line.split()
            out[name] = dict(calls=int(calls), ms=float(ms), work=float(work))
        return out

    def workspace_bytes(self) -> int:
        return 0 if self._engine is None else int(_lib.load().omnitok_engine_workspace_bytes(self._engine))


def load_vqgan(tokenizer, vqgan_ckpt, device=torch.device("cpu"), **kw):
    """reference download.py:48-53: checkpoint -> eval-mode model on `device` (`tokenizer` is the
    reference's unused selector argument)."""
    vqgan = OmniTokenizer_VQGAN.load_from_checkpoint(vqgan_ckpt, strict=False, **kw).to(device)
    vqgan.eval()
    return vqgan


# ------------------------------------------------------------------------------------------------
# encode / decode as PyTorch operators (omnitok::vqgan_encode, omnitok::vqgan_decode): the module's methods go
# through them, and the fake (meta) implementations give torch.compile / torch.export the output shapes.
# Registration failure is an import error: the operators ARE the encode/decode path.
# ------------------------------------------------------------------------------------------------
def _register_ops():
    from torch.library import custom_op

    @custom_op("omnitok::vqgan_encode", mutates_args=(), device_types="cuda")
    def _enc(x: torch.Tensor, handle: int, want_emb: bool, want_z: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        return _model(handle)._encode_native(x.contiguous(), want_emb, want_z)

    @_enc.register_fake
    def _(x, handle, want_emb, want_z):
        m = _model(handle)
        B, C, F, H, W = x.shape
        T, h, w = m.latent_dims(F, H, W)
        ids = x.new_empty((B, T, h, w), dtype=torch.int64)
        if not want_emb:
            emb = x.new_empty(0)
        elif m.use_external_codebook:
            emb = x.new_empty((B, T, h, w, m.cfg.dim))
        else:
            emb = x.new_empty((B, m.cfg.codebook_dim, T, h, w))
        z = x.new_empty((B, T, h, w, m.cfg.codebook_dim)) if want_z else x.new_empty(0)
        return ids, emb, z

    @custom_op("omnitok::vqgan_decode", mutates_args=(), device_types="cuda")
    def _dec(ids: torch.Tensor, handle: int) -> torch.Tensor:
        return _model(handle)._decode_native(ids.contiguous())

    @_dec.register_fake
    def _(ids, handle):
        m = _model(handle)
        B, T, h, w = ids.shape
        F, H, W = m.pixel_dims(T, h, w)
        return ids.new_empty((B, m.cfg.image_channels, F, H, W), dtype=torch.float32)


_register_ops()
