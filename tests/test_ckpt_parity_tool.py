"""CPU: the checkpoint-parity tool (tests/ckpt_parity.py, launched as tools/ckpt_parity.py) on a synthetic PL-format
checkpoint -- load_from_checkpoint's key filtering, the checker legs, the JSON lines.  The GPU legs are covered by
tests/test_gpu_e2e.py::test_ckpt_parity_tool_on_the_gpu."""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from omnitokenizer_amd import make_args, synth
from omnitokenizer_amd.config import OmniTokConfig
from tests import ckpt_parity


def make_ckpt(path, **overrides):
    """what pytorch_lightning writes for the reference (omnitokenizer.py:208 save_hyperparameters, download.py:49)"""
    args = make_args(2, **overrides)
    cfg = OmniTokConfig.from_args(args)
    sd = synth.synth_state_dict(cfg, seed=0)
    sd["image_discriminator.blocks.0.weight"] = torch.zeros(4, 4)     # off-path entries a real checkpoint carries
    sd["perceptual_model.net.slice1.0.weight"] = torch.zeros(3, 3)
    torch.save({"state_dict": sd, "hyper_parameters": {"args": args}, "epoch": 7, "global_step": 1234}, path)
    return args


def test_cpu_only_legs_on_a_synthetic_checkpoint(tmp_path, capsys):
    ck = str(tmp_path / "synthetic.ckpt")
    make_ckpt(ck, resolution=64)
    assert ckpt_parity.main(["--ckpt", ck, "--synthetic", "2", "--cpu-only", "--batch", "2"]) == 0
    lines = [json.loads(l) for l in capsys.readouterr().out.strip().splitlines()]
    setup = lines[0]
    assert setup["event"] == "setup" and setup["off_path_skipped"] == 2 and setup["inputs"] == 2
    assert setup["checker"] in ("reference", "oracle")
    if setup["checker"] == "reference":  # the build container: the oracle is cross-checked against the reference
        x = [l for l in lines if l["event"] == "oracle_vs_reference"]
        assert x and all(l["flips"] == 0 and l["z_err"] < 1e-5 and l["pixel_err"] < 1e-4 for l in x)
    chk = [l for l in lines if l["event"] == "checker"]
    assert chk and all(np.isfinite(l["psnr_recon_vs_input"]) for l in chk)


def test_image_directory_and_clip_inputs(tmp_path, capsys):
    ck = str(tmp_path / "synthetic.ckpt")
    make_ckpt(ck, resolution=64)
    d = tmp_path / "imgs"
    d.mkdir()
    rng = np.random.default_rng(0)
    for i in range(5):
        np.save(d / f"f{i:02d}.npy", (rng.random((80, 96, 3)) * 255).astype(np.float32))  # H, W, C in 0..255
    assert ckpt_parity.main(["--ckpt", ck, "--images", str(d), "--frames", "5", "--cpu-only", "--oracle"]) == 0
    lines = [json.loads(l) for l in capsys.readouterr().out.strip().splitlines()]
    assert lines[0]["inputs"] == 1 and lines[0]["frames"] == 5 and lines[0]["checker"] == "oracle"
    with pytest.raises(SystemExit):
        ckpt_parity.main(["--ckpt", ck, "--images", str(d), "--frames", "4", "--cpu-only"])  # (F - 1) % pt != 0


def test_without_a_gpu_the_product_leg_refuses(tmp_path):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ck = str(tmp_path / "synthetic.ckpt")
    make_ckpt(ck, resolution=64)
    with pytest.raises(SystemExit, match="no CPU fallback"):
        ckpt_parity.main(["--ckpt", ck, "--synthetic", "1", "--oracle"])
