"""CPU: pins the oracle (oracle/omnitok_oracle.py, oracle/vq_argmin.c) against the committed
outputs of the reference itself (tests/golden/*.npz)."""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle import omnitok_oracle as orc
from tests.helpers import (GoldenCase, E2E_CASES, EXT_CASES, FULL_CASES, GOLDEN, HEAVY_BATCH_CASE, HEAVY_CASES, VAE_CASES,
                           VARIANT_CASES)
import os

FAST = [c for c in E2E_CASES if "r256" not in c]


@pytest.mark.parametrize("n_codes", [8192, 16384])
def test_vq_c_oracle_matches_reference_kat(n_codes):
    g = np.load(os.path.join(GOLDEN, f"vq_kat_{n_codes}.npz"))
    ids = c_oracle.vq_argmin(g["z"], g["codebook"])
    assert np.array_equal(ids, g["ids"].astype(np.int64))
    # ties resolve to the lowest index (reference codebook.py:86, torch.argmin)
    assert (ids[4096:4160] == 17).all() and (ids[4160:4200] == 3).all()


@pytest.mark.parametrize("n_codes", [8192])
def test_vq_torch_oracle_matches_reference_kat(n_codes):
    g = np.load(os.path.join(GOLDEN, f"vq_kat_{n_codes}.npz"))
    ids = orc.vq_argmin(torch.from_numpy(g["z"]), torch.from_numpy(g["codebook"])).numpy()
    assert np.array_equal(ids, g["ids"].astype(np.int64))


# (the 17-frame legacy-mode fixture costs ~100 s of einsum attention on the CPU: it is checked against the HIP path
# by tests/test_gpu_e2e.py and by the oracle only in its sdpa twin, keeping this suite at a few minutes)
@pytest.mark.parametrize("name", FAST + ["s2_sdpa_r256_img"] + [c for c in FULL_CASES if "legacy" not in c] + VARIANT_CASES)
def test_oracle_end_to_end_matches_reference(name):
    c = GoldenCase(name)
    with torch.no_grad():
        taps = {}
        ids = orc.encode(c.sd, c.x, c.is_image, c.cfg, taps=taps)
        recon = orc.decode(c.sd, c.ids, c.is_image, c.cfg)
    assert torch.equal(ids, c.ids), f"{(ids != c.ids).sum().item()} ids differ"
    # fp32 noise only: ATen's blocked CPU SDPA (the reference) vs the oracle's explicit softmax; the largest
    # observed is 2.3e-6 (var_up_r_r64_img, 256-token sequences after the Up block), |z| <= 1
    assert (taps["z"] - c.z).abs().max().item() < 4e-6
    assert (c.strided(recon) - c.recon).abs().max().item() < 2e-5
    # flat ids decode identically (reference omnitokenizer.py:272-286)
    if c.is_image or c.cfg.resolution // c.cfg.patch_size == c.ids.shape[-1]:
        recon_flat = orc.decode(c.sd, c.ids.reshape(c.ids.shape[0], -1), c.is_image, c.cfg)
        assert torch.equal(recon_flat, recon)


@pytest.mark.parametrize("name", [c for c in HEAVY_CASES if "r256" not in c])
def test_oracle_matches_reference_on_heavy_statistics(name):
    """The fixtures with trained-checkpoint-like statistics (synth profile "heavy", image-like / constant-colour inputs):
    the oracle still reproduces the reference to fp32 noise, far below the reference's own distance to fp64."""
    c = GoldenCase(name)
    assert c.profile == "heavy" and c.fp32_noise_pix > 0
    with torch.no_grad():
        taps = {}
        ids = orc.encode(c.sd, c.x, c.is_image, c.cfg, taps=taps)
        recon = orc.decode(c.sd, c.ids, c.is_image, c.cfg)
    assert torch.equal(ids, c.ids)
    assert (taps["z"] - c.z).abs().max().item() < 4e-6
    assert (c.strided(recon) - c.recon).abs().max().item() < 1e-4
    assert c.recon_absmax > 5.0  # an order of magnitude beyond the default generator's outputs


def test_oracle_vq_on_golden_z_is_bit_exact():
    for name in FAST:
        c = GoldenCase(name)
        z = c.z.reshape(-1, c.z.shape[-1])
        ids = c_oracle.vq_argmin(z.numpy(), c.sd["codebook.embeddings"].numpy())
        assert np.array_equal(ids, c.ids.reshape(-1).numpy()), name


def test_include_embeddings_matches_reference():
    c = GoldenCase("s2_sdpa_r64_vid")
    with torch.no_grad():
        emb, ids = orc.encode(c.sd, c.x, c.is_image, c.cfg, include_embeddings=True)
    assert torch.equal(ids, c.ids)
    assert (emb.permute(0, 2, 3, 4, 1) - c.emb).abs().max().item() < 1e-6


def test_bias_table_equals_pairwise_bias():
    c = GoldenCase("s1_legacy_r64_img")
    p = "encoder.enc_spatial_transformer.layers.0.1.spatial_rel_pos_bias"
    h = w = 8
    full = orc.continuous_position_bias(c.sd, p, h, w)
    tab = orc.continuous_position_bias_table(c.sd, p, h, w)
    ys, xs = np.divmod(np.arange(h * w), w)
    dy = ys[:, None] - ys[None, :] + h - 1
    dx = xs[:, None] - xs[None, :] + w - 1
    assert torch.equal(full, tab[:, dy, dx])


@pytest.mark.parametrize("name", [c for c in VAE_CASES if "r256" not in c])
def test_oracle_vae_matches_reference(name):
    """--use_vae path (reference omnitokenizer.py:260-266, 293-317) with the reference's own noise."""
    c = GoldenCase(name)
    with torch.no_grad():
        z = orc.encode_vae(c.sd, c.x, c.is_image, c.cfg, noise=c.noise)
        # the reference's host draw is reproducible from the seed alone
        torch.manual_seed(c.noise_seed)
        z_seeded = orc.encode_vae(c.sd, c.x, c.is_image, c.cfg)
        recon = orc.decode_vae(c.sd, c.decode_input(), c.is_image, c.cfg)
        flat = c.z.permute(0, 2, 3, 4, 1).reshape(c.z.shape[0], -1, c.z.shape[1])
        recon_flat = orc.decode_vae(c.sd, flat, c.is_image, c.cfg)
    z5 = z.unsqueeze(2) if c.is_image else z
    assert tuple(z5.shape) == tuple(c.z.shape)
    assert (z5 - c.z).abs().max().item() < 1e-5
    assert torch.equal(z, z_seeded)
    assert (c.strided(recon) - c.recon).abs().max().item() < 2e-5
    assert torch.equal(recon, recon_flat)


def test_vq_cos_c_oracle_matches_reference_kat():
    g = np.load(os.path.join(GOLDEN, "vq_cos_kat_8192.npz"))
    ids = c_oracle.vq_argmax_cos(g["z"], g["codebook"])
    assert np.array_equal(ids, g["ids"].astype(np.int64))
    assert (ids[3000:3064] == 17).all() and (ids[3064:3100] == 3).all()  # ties -> lowest index
    t = orc.vq_argmax_cos(torch.from_numpy(g["z"]), torch.from_numpy(g["codebook"])).numpy()
    assert np.array_equal(t, ids)


def test_vq_cdist_c_oracle_matches_reference_kat():
    g = np.load(os.path.join(GOLDEN, "vq_cdist_kat_8192.npz"))
    ids = c_oracle.vq_argmin_cdist(g["z"], g["codebook"])
    assert np.array_equal(ids, g["ids"].astype(np.int64))
    assert (ids[3000:3064] == 17).all() and (ids[3064:3100] == 3).all()
    t = orc.vq_argmin_cdist(torch.from_numpy(g["z"]), torch.from_numpy(g["codebook"])).numpy()
    assert np.array_equal(t, ids)


@pytest.mark.parametrize("name", [c for c in EXT_CASES if "r128" not in c])
def test_oracle_external_codebook_matches_reference(name):
    """--use_external_codebook (VectorQuantize, cosine similarity)."""
    c = GoldenCase(name)
    with torch.no_grad():
        taps = {}
        emb, ids = orc.encode(c.sd, c.x, c.is_image, c.cfg, include_embeddings=True, taps=taps)
        recon = orc.decode(c.sd, c.ids, c.is_image, c.cfg)
    assert torch.equal(ids, c.ids)
    assert (taps["z"] - c.z).abs().max().item() < 2e-6
    assert (emb.permute(0, 2, 3, 4, 1)[..., ::8] - c.emb).abs().max().item() < 1e-6
    assert (c.strided(recon) - c.recon).abs().max().item() < 2e-5
    cq = c_oracle.vq_argmax_cos if c.cfg.l2_code else c_oracle.vq_argmin_cdist
    cids = cq(c.z.reshape(-1, 8).numpy(), c.sd["codebook._codebook.embed"][0].numpy())
    assert np.array_equal(cids, c.ids.reshape(-1).numpy())


def test_heavy_batch_fixture_is_consistent_and_pins_the_oracle():
    """The batch-scale heavy fixture (the reference in fp32 AND fp64 on 8 clips): internal consistency of what the GPU test
    uses as its yardstick, and the oracle against the reference on one of its clips (fp32, and in fp64 against the stored
    fp64 latents)."""
    c = GoldenCase(HEAVY_BATCH_CASE)
    n = c.ids.numel()
    assert n == 8 * 5 * 32 * 32 and c.boundary.numel() == n and c.ref_flips == int((c.ids != c.ids64).sum())
    E = c.sd["codebook.embeddings"].double()
    z64 = c.z64.reshape(-1, 8)
    # the stored fp64 ids ARE the nearest codes of the stored fp64 latents, and the boundary distances are positive
    d = (z64 * z64).sum(1, keepdim=True)[:4096] - 2.0 * z64[:4096] @ E.t() + (E * E).sum(1)[None]
    assert torch.equal(d.argmin(1), c.ids64.reshape(-1)[:4096])
    assert float(c.boundary.min()) > 0.0
    ref_l2 = (c.z.reshape(-1, 8).double() - z64).norm(dim=1)
    assert abs(float(ref_l2.max()) - c.fp32_noise_l2_max) < 1e-9
    # the reference's own fp32 run flips nothing: its per-token error is inside every token's cell
    assert int((c.boundary < ref_l2).sum()) == 0 and c.ref_flips == 0
    b = 3
    x = c.x[b:b + 1]
    with torch.no_grad():
        taps = {}
        ids = orc.encode(c.sd, x, False, c.cfg, taps=taps)
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in c.sd.items()}
        taps64 = {}
        ids64 = orc.encode(sd64, x.double(), False, c.cfg, taps=taps64)
    assert torch.equal(ids, c.ids[b:b + 1]) and torch.equal(ids64, c.ids64[b:b + 1])
    assert (taps["z"] - c.z[b:b + 1]).abs().max().item() < 2e-5          # fp32 vs fp32: summation order only
    assert (taps64["z"] - c.z64[b:b + 1]).abs().max().item() < 2e-6      # fp64 vs fp64 (RoPE table precision)


def test_codebook_usage_state_matches_reference():
    """Reference Codebook.forward mutates `codebook_usage` / `call_cnt` on every call, also in eval mode, i.e. on every
    encode() (codebook.py:122-143): the oracle's restatement reproduces the reference's buffer after three encodes
    (fixture made by tests/golden/make_golden.py::make_usage_state_golden from the reference itself)."""
    g = np.load(os.path.join(GOLDEN, "usage_state_s2_sdpa_r64.npz"))
    usage, cnt = torch.zeros(8192), 0
    for i in range(3):
        ids = torch.from_numpy(g[f"ids{i}"].astype(np.int64))
        _, _, _, usage, cnt = orc.codebook_usage_update(ids, 8192, usage, cnt)
        assert (usage - torch.from_numpy(g["usage"][i])).abs().max().item() < 1e-8
    assert cnt == int(g["call_cnt"]) == 3
