"""omnitokenizer_amd -- MI355X-native (gfx950 HIP) encode/decode path of OmniTokenizer.

    from omnitokenizer_amd import OmniTokenizer_VQGAN     # drop-in for the reference class
    from omnitokenizer_amd.gpt import GPT, sample_with_past, sample_with_past_cfg   # LM consumer
"""
from .config import OmniTokConfig, make_args  # noqa: F401

__all__ = ["OmniTokenizer_VQGAN", "GPT", "OmniTokConfig", "make_args"]


def __getattr__(name):
    if name == "OmniTokenizer_VQGAN":
        from .vqgan import OmniTokenizer_VQGAN
        return OmniTokenizer_VQGAN
    if name == "GPT":
        from .gpt import GPT
        return GPT
    raise AttributeError(name)
