"""The tokenizer path as C++-registered PyTorch operators (csrc/torch_binding.cpp, lib/libomnitok_torch.so):

    torch.classes.omnitok.Engine                      the native engine as a script object (torch::CustomClassHolder)
    torch.ops.omnitok.engine_encode(e, x) -> ids      VQGAN.encode  (reference omnitokenizer.py:247-266)
    torch.ops.omnitok.engine_encode_full(e, x) -> (ids, embeddings, z)
    torch.ops.omnitok.engine_decode(e, ids) -> pixels VQGAN.decode  (reference omnitokenizer.py:293-317)

Schemas are tensor-only plus the engine object: a program captured with torch.export carries the engine as a constant
and runs wherever `torch.ops.load_library(LIB_PATH)` was called -- no Python-side handle table.  This module adds what
TRACING needs and C++ cannot give: the fake class of the script object and the operators' fake implementations (shapes
from the engine's own host-side shape functions), plus `engine_from_module` / `EngineModule` to build an engine from an
`OmniTokenizer_VQGAN`'s configuration and state_dict.
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

import torch
from torch import nn

from . import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libomnitok_torch.so")
_loaded = False


def load() -> None:
    """Loads libomnitok_torch.so (built by `python omnitokenizer_amd/build.py`) and registers the tracing side.  A missing
    library is an error: the operators have no Python implementation."""
    global _loaded
    if _loaded:
        return
    _lib.load()  # libomnitok.so first (it must see PyTorch's HIP runtime, see _lib.load)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python omnitokenizer_amd/build.py` (it compiles the TORCH_LIBRARY "
                           "binding with g++ against this PyTorch)")
    torch.ops.load_library(LIB_PATH)
    _register_tracing()
    _loaded = True


def _register_tracing() -> None:
    from torch._library.fake_class_registry import register_fake_class

    @register_fake_class("omnitok::Engine")
    class FakeEngine:  # noqa: F841 -- registered by the decorator
        """What fake-tensor tracing sees instead of the engine: the configuration, and a weight-less native engine
        (omnitok_engine_create is host-only) that answers the shape questions."""

        def __init__(self, cfg: Dict[str, int], enc_block: str, dec_block: str):
            self.cfg, self.enc_block, self.dec_block = dict(cfg), enc_block, dec_block
            self._shapes = torch.classes.omnitok.Engine(self.cfg, enc_block, dec_block)

        @classmethod
        def __obj_unflatten__(cls, flat):
            d = dict(flat)
            return cls(d["cfg"], d["enc_block"], d["dec_block"])

        def encode_shape(self, F: int, H: int, W: int):
            return self._shapes.encode_shape(F, H, W)

        def decode_shape(self, T: int, h: int, w: int):
            return self._shapes.decode_shape(T, h, w)

        def config(self):
            return self.cfg

        def blocks(self):
            return [self.enc_block, self.dec_block]

        def _not_traceable(self, *a, **k):
            raise RuntimeError("omnitok.Engine: weights and options are set before tracing, not inside the traced program")

        set_weight = finalize = missing = set_option = _not_traceable

    def _video_dims(x):
        if x.dim() == 4:
            return x.shape[0], 1, x.shape[2], x.shape[3]
        if x.dim() != 5:
            raise ValueError("omnitok engine_encode: x must be [B,C,F,H,W] or [B,C,H,W]")
        return x.shape[0], x.shape[2], x.shape[3], x.shape[4]

    @torch.library.register_fake("omnitok::engine_encode")
    def _(e, x):
        B, F, H, W = _video_dims(x)
        T, h, w = e.encode_shape(int(F), int(H), int(W))
        return x.new_empty((B, T, h, w), dtype=torch.int64)

    @torch.library.register_fake("omnitok::engine_encode_full")
    def _(e, x):
        B, F, H, W = _video_dims(x)
        T, h, w = e.encode_shape(int(F), int(H), int(W))
        c = e.config()
        emb = x.new_empty((B, T, h, w, c["dim"])) if c.get("external_codebook", 0) else x.new_empty((B, c["codebook_dim"], T, h, w))
        return x.new_empty((B, T, h, w), dtype=torch.int64), emb, x.new_empty((B, T, h, w, c["codebook_dim"]))

    @torch.library.register_fake("omnitok::engine_decode")
    def _(e, ids):
        B, T, h, w = ids.shape
        F, H, W = e.decode_shape(int(T), int(h), int(w))
        return ids.new_empty((B, e.config()["image_channels"], F, H, W), dtype=torch.float32)


def native_config_dict(vqgan) -> Tuple[Dict[str, int], str, str]:
    """omnitok_config of an OmniTokenizer_VQGAN as (int fields, enc_block, dec_block)."""
    nc = vqgan._native_config()
    ints = {k: int(getattr(nc, k)) for k, _ in nc._fields_ if k not in ("enc_block", "dec_block")}
    return ints, nc.enc_block.decode(), nc.dec_block.decode()


def engine_from_module(vqgan):
    """A torch.classes.omnitok.Engine with the configuration and (GPU-resident) weights of `vqgan`."""
    load()
    ints, enc, dec = native_config_dict(vqgan)
    e = torch.classes.omnitok.Engine(ints, enc, dec)
    anchor = None
    for name, t in vqgan.state_dict(keep_vars=True).items():
        t = t.detach()
        if t.device.type != "cuda":
            raise RuntimeError(f"{name} is on {t.device}: move the module to the GPU first (there is no CPU path)")
        if t.dtype.is_floating_point and t.dtype != torch.float32:
            raise TypeError(f"{name}: parameters must be float32 (the path computes in fp32 like the reference)")
        e.set_weight(name, t)
        anchor = t if anchor is None else anchor
    if anchor is None:
        raise RuntimeError("the module has no parameters")
    e.finalize(anchor)
    return e


class EngineModule(nn.Module):
    """encode -> ids, decode -> pixels over the C++-registered operators; exportable with torch.export."""

    def __init__(self, engine):
        super().__init__()
        self.engine = engine

    def encode(self, x: torch.Tensor) -> torch.Tensor:
        return torch.ops.omnitok.engine_encode(self.engine, x)

    def decode(self, ids: torch.Tensor) -> torch.Tensor:
        return torch.ops.omnitok.engine_decode(self.engine, ids)

    def forward(self, x: torch.Tensor):
        ids = self.encode(x)
        return self.decode(ids), ids
