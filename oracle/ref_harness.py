"""TEST INFRASTRUCTURE ONLY -- imports the *unmodified* reference (read-only, from
/root/reference) behind stubs for its non-arithmetic dependencies, so that golden
vectors can be generated in the build container.

/root/reference does not exist on the GPU box: nothing in `-m gpu` tests, smoke() or
bench.py imports this module.  It is used by tests/golden/make_golden.py (fixture
generation) and by CPU tests that are skipped when the reference is absent.

Only packages that carry no arithmetic on the encode/decode path are stubbed:
pytorch_lightning (LightningModule -> nn.Module), timm (trunc_normal_ -> torch's own),
beartype (identity decorator), fairscale, imageio, torchvision, LPIPS (a loss net).
"""
import importlib
import os
import sys
import types
import contextlib

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("OMNITOK_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "OmniTokenizer"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_installed = False


def install_stubs():
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

        @property
        def device(self):
            return next(self.parameters()).device

    _mod("pytorch_lightning", LightningModule=LightningModule)

    class DropPath(nn.Identity):
        pass

    def to_2tuple(x):
        return (x, x) if not isinstance(x, tuple) else x

    _mod("timm")
    _mod("timm.models")
    _mod("timm.models.layers", trunc_normal_=nn.init.trunc_normal_, DropPath=DropPath,
         to_2tuple=to_2tuple)
    _mod("timm.scheduler")
    _mod("timm.scheduler.cosine_lr", CosineLRScheduler=object)
    _mod("beartype", beartype=lambda f: f)
    _mod("fairscale")
    _mod("fairscale.nn", checkpoint_wrapper=lambda m, *a, **k: m)
    _mod("imageio")
    tv = _mod("torchvision")
    tv.models = _mod("torchvision.models")
    tv.transforms = _mod("torchvision.transforms")

    pkg = types.ModuleType("OmniTokenizer")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "OmniTokenizer")]
    sys.modules["OmniTokenizer"] = pkg
    mods = types.ModuleType("OmniTokenizer.modules")
    mods.__path__ = [os.path.join(REFERENCE_ROOT, "OmniTokenizer", "modules")]
    sys.modules["OmniTokenizer.modules"] = mods

    class LPIPS(nn.Module):  # perceptual loss net: training only, not on the path
        def forward(self, *a, **k):
            raise RuntimeError("LPIPS stub")

    _mod("OmniTokenizer.modules.lpips", LPIPS=LPIPS)
    mods.LPIPS = LPIPS
    cb = importlib.import_module("OmniTokenizer.modules.codebook")
    mods.Codebook = cb.Codebook
    disc = importlib.import_module("OmniTokenizer.modules.discriminator")  # pure torch
    mods.ApplyNoise, mods.ApplyStyle, mods.Blur2d = disc.ApplyNoise, disc.ApplyStyle, disc.Blur2d
    _installed = True


def load_reference():
    """Returns the reference module OmniTokenizer.omnitokenizer (class VQGAN)."""
    install_stubs()
    return importlib.import_module("OmniTokenizer.omnitokenizer")


def build_reference_model(args):
    """Reference VQGAN(args).eval() on CPU with codebook init disabled
    (vqgan_eval.py:80 sets _need_init=False the same way)."""
    ref = load_reference()
    torch.manual_seed(0)
    model = ref.VQGAN(args).eval()
    model.codebook._need_init = False
    return model


@contextlib.contextmanager
def attention_mode(mode: str):
    """'sdpa' = the branch torch>=2.1 takes; 'legacy' = the einsum branch, selected the
    way SURVEY.md A.1-Q2 describes: the reference tests torch.__version__ as a string
    (attention.py:439)."""
    assert mode in ("sdpa", "legacy")
    old = torch.__version__
    try:
        if mode == "legacy":
            torch.__version__ = "2.0.0"
        yield
    finally:
        torch.__version__ = old
