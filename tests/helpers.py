"""Shared helpers for the parity tests: rebuild a golden case's config/weights/inputs from the
seeds stored in the fixture (tests/golden/make_golden.py wrote them)."""
import ast
import os
import zlib

import numpy as np
import torch

from omnitokenizer_amd import synth
from omnitokenizer_amd.config import OmniTokConfig, make_args

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

E2E_CASES = ["s2_sdpa_r64_img", "s2_sdpa_r64_vid", "s2_sdpa_r64_vid9", "s1_legacy_r64_img",
             "s1_legacy_r64_vid", "s1_sdpa_r64_img", "s2_sdpa_r128_vid_16k", "s2_sdpa_r256_img",
             "s2_sdpa_r256_vid", "s1_legacy_r256_img"]
SMALL_CASES = [c for c in E2E_CASES if "r64" in c]
# the headline configuration at full length: one 17-frame 256x256 clip of the stage-2 (C3) and of the
# stage-1 / legacy-attention (C1-style) architectures
FULL_CASES = ["s2_sdpa_r256_vid17", "s1_legacy_r256_vid17"]


VARIANT_CASES = ["var_pool_a_r128_vid", "var_pool_m_r128_img", "var_pool_l_r128_vid", "var_cnn_r128_img",
                 "var_cnn_r128_vid", "var_defer_t_r128_vid", "var_defer_s_r128_vid", "var_defer_ts_r128_img",
                 "var_genup2_r64_vid", "var_up_n_r64_vid", "var_up_r_r64_img", "var_up_r_pool_r128_vid"]
# external VectorQuantize: cosine-similarity codebook (l2_code) and Euclidean codebook (no l2_code)
EXT_CASES = ["ext_s2_sdpa_r64_img", "ext_s2_sdpa_r64_vid", "ext_s1_legacy_r128_vid", "ext_euclid_s2_sdpa_r64_img",
             "ext_euclid_s2_sdpa_r64_vid"]
# trained-checkpoint-like statistics (synth profile "heavy") and image-like / constant-colour inputs
HEAVY_CASES = ["heavy_s2_sdpa_r64_img", "heavy_s2_sdpa_r64_vid", "heavy_s1_legacy_r64_vid", "heavy_s2_sdpa_r128_vid_16k",
               "heavy_s2_sdpa_r256_vid17"]
# 8 distinct 17x256x256 clips, heavy profile, the reference in fp32 AND fp64 (ids, z) -- 40 960 tokens
HEAVY_BATCH_CASE = "heavy_s2_sdpa_r256_vid17_b8"
VAE_CASES = ["vae_s2_sdpa_r64_img", "vae_s2_sdpa_r64_vid", "vae_s1_legacy_r64_vid", "vae_s2_sdpa_r256_vid"]


class GoldenCase:
    def __init__(self, name):
        g = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.name = name
        self.stage = int(g["stage"])
        self.mode = str(g["mode"])
        self.overrides = ast.literal_eval(str(g["overrides"]))
        self.batch = int(g["batch"])
        self.frames = int(g["frames"])
        self.stride = int(g["stride"])
        self.is_image = self.frames == 1
        self.args = make_args(self.stage, **self.overrides)
        self.cfg = OmniTokConfig.from_args(self.args, attention_mode=self.mode)
        self.profile = str(g["profile"]) if "profile" in g.files else "default"
        self.input_kind = str(g["input_kind"]) if "input_kind" in g.files else "noise"
        self.sd = synth.synth_state_dict(self.cfg, seed=int(g["weight_seed"]), profile=self.profile)
        assert synth.state_checksum(self.sd) == int(g["state_crc"]), \
            "synthetic weight generator drifted from the golden fixtures"
        res = self.cfg.resolution
        seed = int(g["input_seed"])
        self.x = (synth.synth_image(self.batch, res, seed, kind=self.input_kind) if self.is_image
                  else synth.synth_video(self.batch, self.frames, res, seed, kind=self.input_kind))
        assert zlib.crc32(self.x.numpy().tobytes()) == int(g["input_crc"]), "synthetic input drifted"
        self.is_vae = "noise" in g.files
        if self.is_vae:  # --use_vae fixture (make_golden.run_vae_case)
            self.noise = torch.from_numpy(g["noise"])      # b c t h w
            self.moments = torch.from_numpy(g["moments"])  # b 2c t h w (mean | raw logvar)
            self.z = torch.from_numpy(g["z"])              # b c t h w posterior sample
            self.noise_seed = int(g["noise_seed"])
        else:
            self.ids = torch.from_numpy(g["ids"].astype(np.int64))
            self.z = torch.from_numpy(g["z"])          # b t h w c
            self.emb = torch.from_numpy(g["emb"])      # b t h w c
        # how far the reference's own fp32 result is from the fp64 one (heavy-statistics fixtures only)
        self.fp32_noise_z = float(g["fp32_noise_z"]) if "fp32_noise_z" in g.files else 0.0
        self.fp32_noise_pix = float(g["fp32_noise_pix"]) if "fp32_noise_pix" in g.files else 0.0
        # batch-scale heavy fixture (make_golden.run_heavy_batch_case): the reference's fp64 run beside its fp32 run
        if "ids64" in g.files:
            self.ids64 = torch.from_numpy(g["ids64"].astype(np.int64))
            self.z64 = self.z.double() + torch.from_numpy(g["z64_resid"]).double()
            self.fp32_noise_z_clip = torch.from_numpy(g["fp32_noise_z_clip"])
            self.fp32_noise_pix_clip = torch.from_numpy(g["fp32_noise_pix_clip"])
            self.fp32_noise_l2_max = float(g["fp32_noise_l2_max"])
            self.ref_flips = int(g["ref_flips"])
            # L2 distance (fp64) of every latent to the nearest boundary of its code's cell: a perturbation smaller than
            # this cannot flip the token
            self.boundary = torch.from_numpy(g["boundary"]).double()
        self.recon = torch.from_numpy(g["recon"])  # strided
        self.perplexity = float(g["perplexity"]) if "perplexity" in g.files else None
        self.recon_absmax = float(g["recon_absmax"])

    def decode_input(self, z5=None):
        """z in the layout reference decode() accepts: image [b,c,h,w], video [b,t,h,w,c]."""
        z5 = self.z if z5 is None else z5
        return z5[:, :, 0] if self.is_image else z5.permute(0, 2, 3, 4, 1).contiguous()

    def strided(self, recon_full):
        s = self.stride
        return recon_full[..., ::s, ::s]
