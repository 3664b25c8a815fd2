"""CPU, build container only (skipped where /root/reference is absent, e.g. the GPU box):
runs the reference live next to the oracle; also checks the synthetic state_dict covers exactly
the reference's path keys."""
import numpy as np
import pytest
import torch

from oracle import ref_harness as rh
from oracle import omnitok_oracle as orc
from oracle import c_oracle
from omnitokenizer_amd import synth
from omnitokenizer_amd.config import OmniTokConfig, make_args

pytestmark = pytest.mark.skipif(not rh.reference_available(), reason="reference not mounted")

OFF_PATH = ("image_discriminator", "video_discriminator", "perceptual_model")


@pytest.mark.parametrize("stage,mode", [(2, "sdpa"), (1, "legacy")])
def test_live_reference_vs_oracle(stage, mode):
    args = make_args(stage, resolution=64)
    cfg = OmniTokConfig.from_args(args, attention_mode=mode)
    sd = synth.synth_state_dict(cfg, seed=3)
    model = rh.build_reference_model(args)
    ref_keys = {k for k in model.state_dict() if not k.startswith(OFF_PATH)}
    assert ref_keys == set(sd.keys())
    for k, v in model.state_dict().items():
        if k in sd:
            assert tuple(v.shape) == tuple(sd[k].shape) and v.dtype == sd[k].dtype, k
    assert not model.load_state_dict(sd, strict=False).unexpected_keys
    x = synth.synth_video(1, 9, 64, seed=5)
    with torch.no_grad(), rh.attention_mode(mode):
        ids_ref = model.encode(x, False)
        rec_ref = model.decode(ids_ref, False)
    with torch.no_grad():
        ids = orc.encode(sd, x, False, cfg)
        rec = orc.decode(sd, ids_ref, False, cfg)
    assert torch.equal(ids, ids_ref)
    assert (rec - rec_ref).abs().max().item() < 2e-5


def test_reference_codebook_vs_c_oracle_random():
    rh.install_stubs()
    from OmniTokenizer.modules.codebook import Codebook
    rng = np.random.default_rng(11)
    E = rng.standard_normal((8192, 8), dtype=np.float32)
    z = rng.standard_normal((16384, 8), dtype=np.float32)
    z /= np.linalg.norm(z, axis=1, keepdims=True)
    cb = Codebook(8192, 8).eval()
    cb._need_init = False
    cb.embeddings.data.copy_(torch.from_numpy(E))
    zt = torch.from_numpy(z).reshape(16, 1, 32, 32, 8).permute(0, 4, 1, 2, 3).contiguous()
    with torch.no_grad():
        ids_ref = cb(zt)["encodings"].reshape(-1).numpy()
    assert np.array_equal(c_oracle.vq_argmin(z, E), ids_ref)
