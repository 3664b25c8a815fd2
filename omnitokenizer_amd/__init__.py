"""omnitokenizer_amd -- MI355X-native (gfx950 HIP) encode/decode path of OmniTokenizer.

    from omnitokenizer_amd import OmniTokenizer_VQGAN     # drop-in for the reference class
"""
from .config import OmniTokConfig, make_args  # noqa: F401

__all__ = ["OmniTokenizer_VQGAN", "OmniTokConfig", "make_args"]


def __getattr__(name):
    if name == "OmniTokenizer_VQGAN":
        from .vqgan import OmniTokenizer_VQGAN
        return OmniTokenizer_VQGAN
    raise AttributeError(name)
