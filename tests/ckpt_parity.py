#!/usr/bin/env python
"""TEST INFRASTRUCTURE (lives under tests/ because it drives the checker in oracle/; `python tools/ckpt_parity.py ...` is a
launcher for this file).  Parity of the MI355X path against the reference ON A TRAINED CHECKPOINT, in one command.

    python tools/ckpt_parity.py --ckpt imagenet_k600.ckpt --images DIR [--frames 17] [--limit 16] [--batch 8]
    python tools/ckpt_parity.py --ckpt X.ckpt --synthetic 4            # no data at hand: synthetic "natural" inputs

What it does (reference README.md:44-61 lists the released checkpoints, OmniTokenizer/download.py:48-53 loads them):
  1. loads the PL-format checkpoint {"state_dict", "hyper_parameters": {"args"}} through
     OmniTokenizer_VQGAN.load_from_checkpoint (the drop-in's own loader; discriminator / LPIPS entries are skipped);
  2. builds the checker from the SAME state_dict: the unmodified reference through oracle/ref_harness.py when
     /root/reference (or $OMNITOK_REFERENCE_ROOT) is mounted, else the CPU oracle (which is pinned to the reference by
     tests/golden) -- `checker` on every output line says which;
  3. per arithmetic mode (default fp16-split planes; strict fp32-MFMA) runs encode / decode on the GPU and prints one
     JSON line per batch and a summary: id flips against the checker, the near-tie audit of every flip (fp64 distance gap of
     the two candidate codes, relative), pre-VQ latent error, decode(checker ids) pixel error, PSNR(ours, checker) and
     PSNR(reconstruction, input) as in evaluation/common_metrics_on_video_quality/calculate_psnr.py:6-15.
Images: every *.png / *.jpg / *.jpeg / *.npy under DIR (sorted), centre-cropped and resized to the checkpoint's resolution,
scaled to [-0.5, 0.5] (reference data.py preprocess contract).  --frames F > 1 stacks F consecutive images into a clip.
The checker runs on the CPU: budget ~1 s per image, ~10 s per 17-frame clip per core-set.

Exit status: 0 when every flip is a provable near-tie (gap < 1e-5) and the pixel error is within --pixel-tol (1e-4 x the
checker's own output range), 1 otherwise.  --cpu-only runs steps 1-2 only (checker vs oracle cross-check; no GPU needed).
"""
import argparse
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def load_inputs(a, res, channels):
    """-> list of [C, F, H, W] float32 tensors in [-0.5, 0.5]"""
    from omnitokenizer_amd import synth
    F = a.frames
    if a.synthetic:
        x = synth.synth_image(a.synthetic, res, seed=1234, kind="natural") if F == 1 else \
            synth.synth_video(a.synthetic, F, res, seed=1234, kind="natural")
        return [t[:, None] if F == 1 else t for t in x]
    paths = sorted(p for ext in ("png", "jpg", "jpeg", "JPEG", "npy") for p in glob.glob(os.path.join(a.images, "**", "*." + ext),
                                                                                   recursive=True))
    if not paths:
        raise SystemExit(f"no images under {a.images}")
    frames = []
    for p in paths[: a.limit * F]:
        if p.endswith(".npy"):
            im = torch.from_numpy(np.load(p)).float()      # [H, W, C] or [C, H, W], 0..255 or 0..1
            if im.dim() == 3 and im.shape[-1] in (1, 3):
                im = im.permute(2, 0, 1)
            if im.max() > 1.5:
                im = im / 255.0
        else:
            from PIL import Image
            im = torch.from_numpy(np.asarray(Image.open(p).convert("RGB"), dtype=np.float32) / 255.0).permute(2, 0, 1)
        c, h, w = im.shape
        s = min(h, w)
        im = im[:, (h - s) // 2:(h - s) // 2 + s, (w - s) // 2:(w - s) // 2 + s]
        im = torch.nn.functional.interpolate(im[None], size=(res, res), mode="bilinear", antialias=True, align_corners=False)[0]
        frames.append(im[:channels] - 0.5)
    n = len(frames) // F
    if n == 0:
        raise SystemExit(f"{len(frames)} images < one clip of {F} frames")
    return [torch.stack(frames[i * F:(i + 1) * F], 1) for i in range(n)]


def psnr(a, b):
    """calculate_psnr.py:6-15 on clamp(x + 0.5, 0, 1); MSE floor 1e-10 -> 100 dB"""
    a, b = (a + 0.5).clamp(0, 1).double(), (b + 0.5).clamp(0, 1).double()
    mse = ((a - b) ** 2).mean().item()
    return 100.0 if mse < 1e-10 else 20 * np.log10(1.0 / np.sqrt(mse))


class Checker:
    """the reference itself when mounted, else the oracle; same state_dict, CPU fp32"""

    def __init__(self, args, sd, mode, force_oracle=False):
        from oracle import ref_harness as rh
        from omnitokenizer_amd.config import OmniTokConfig
        self.mode = mode
        self.cfg = OmniTokConfig.from_args(args, attention_mode=mode)
        self.sd = {k: v.detach().cpu() for k, v in sd.items()}
        self.ref = None
        if rh.reference_available() and not force_oracle:
            self.rh = rh
            self.ref = rh.build_reference_model(args)
            msg = self.ref.load_state_dict(self.sd, strict=False)
            if msg.unexpected_keys:
                print(f"# reference ignored {len(msg.unexpected_keys)} keys", file=sys.stderr)
        self.kind = "reference" if self.ref is not None else "oracle"

    @torch.no_grad()
    def run(self, x, is_image):
        """x [B, C, F, H, W] -> ids [B,T,h,w], z [B,T,h,w,c], recon"""
        xin = x[:, :, 0] if is_image else x
        if self.ref is not None:
            with self.rh.attention_mode(self.mode):
                ids = self.ref.encode(xin, is_image)
                h = self.ref.pre_vq_conv(self.ref.encoder(xin, is_image))
                z = torch.nn.functional.normalize(h, p=2, dim=1) if self.cfg.l2_code else h
                rec = self.ref.decode(ids, is_image)
            return ids, z.permute(0, 2, 3, 4, 1).contiguous(), rec
        from oracle import omnitok_oracle as orc
        taps = {}
        ids = orc.encode(self.sd, xin, is_image, self.cfg, taps=taps)
        return ids, taps["z"], orc.decode(self.sd, ids, is_image, self.cfg)


def near_tie_audit(ids, ids_ref, z, E):
    """every differing id -> relative fp64 gap between the two candidate codes' distances to OUR latent"""
    ids, ids_ref = ids.reshape(-1).cpu(), ids_ref.reshape(-1).cpu()
    bad = (ids != ids_ref).nonzero().flatten()
    if bad.numel() == 0:
        return 0, []
    zz = z.reshape(-1, z.shape[-1]).cpu().double()[bad]
    E = E.double().cpu()
    d_o = ((zz - E[ids[bad]]) ** 2).sum(1)
    d_r = ((zz - E[ids_ref[bad]]) ** 2).sum(1)
    return int(bad.numel()), ((d_o - d_r).abs() / d_r.clamp_min(1e-12)).tolist()


MODES = {"default": dict(gemm_mode=2, attn_mode=1), "strict_fp32": dict(gemm_mode=0, attn_mode=0)}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--ckpt", required=True)
    ap.add_argument("--images", help="directory of images (png / jpg / npy)")
    ap.add_argument("--synthetic", type=int, default=0, help="N synthetic image-like inputs instead of --images")
    ap.add_argument("--frames", type=int, default=1, help="frames per clip (1 = images)")
    ap.add_argument("--limit", type=int, default=16, help="clips / images to check")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--attention-mode", default=None, help="sdpa | legacy (imagenet_only.ckpt: legacy, README.md:58)")
    ap.add_argument("--modes", default="default,strict_fp32")
    ap.add_argument("--pixel-tol", type=float, default=1e-4)
    ap.add_argument("--oracle", action="store_true", help="use the CPU oracle as the checker even when the reference is mounted")
    ap.add_argument("--trust-checkpoint", action="store_true",
                    help="fall back to a full unpickle when the file does not load with weights_only=True (runs the file's code)")
    ap.add_argument("--cpu-only", action="store_true", help="load + checker legs only (cross-checks reference vs oracle when both exist)")
    a = ap.parse_args(argv)
    if not a.images and not a.synthetic:
        ap.error("--images DIR or --synthetic N")

    from omnitokenizer_amd.vqgan import load_checkpoint_file
    ckpt = load_checkpoint_file(a.ckpt, trust=a.trust_checkpoint)
    hp = ckpt["hyper_parameters"]["args"]
    mode = a.attention_mode or getattr(hp, "attention_mode", "sdpa")
    from omnitokenizer_amd import OmniTokenizer_VQGAN
    model = OmniTokenizer_VQGAN.load_from_checkpoint(a.ckpt, strict=False, attention_mode=mode, trust_checkpoint=a.trust_checkpoint)
    cfg = model.cfg
    pt = cfg.enc_temporal_patch_size
    if a.frames > 1 and (a.frames - 1) % pt:
        raise SystemExit(f"--frames {a.frames}: (frames - 1) must be divisible by temporal_patch_size {pt}")
    is_image = a.frames == 1
    sd = {k: v for k, v in model.state_dict().items()}
    chk = Checker(hp, sd, mode, force_oracle=a.oracle)
    clips = load_inputs(a, cfg.resolution, cfg.image_channels)[: a.limit]
    print(json.dumps(dict(event="setup", ckpt=os.path.basename(a.ckpt), checker=chk.kind, attention_mode=mode, inputs=len(clips),
                          frames=a.frames, resolution=cfg.resolution, n_codes=cfg.n_codes,
                          tensors_loaded=len(sd), off_path_skipped=len(ckpt["state_dict"]) - len(sd))))
    E = sd.get("codebook.embeddings")
    ok = True
    results = {}
    batches = [torch.stack(clips[i:i + a.batch]) for i in range(0, len(clips), a.batch)]
    refs = [chk.run(x, is_image) for x in batches]  # CPU, once
    if a.cpu_only or not torch.cuda.is_available():
        if not a.cpu_only:
            raise SystemExit("no GPU visible: the MI355X path has no CPU fallback (use --cpu-only for the checker legs alone)")
        if chk.kind == "reference":  # cross-check the two CPU checkers on this checkpoint
            orc_chk = Checker(hp, sd, mode, force_oracle=True)
            for bi, x in enumerate(batches):
                ids_o, z_o, rec_o = orc_chk.run(x, is_image)
                flips, gaps = near_tie_audit(ids_o, refs[bi][0], z_o, E)
                print(json.dumps(dict(event="oracle_vs_reference", batch=bi, tokens=ids_o.numel(), flips=flips, gaps=gaps,
                                      z_err=(z_o - refs[bi][1]).abs().max().item(),
                                      pixel_err=(rec_o - refs[bi][2]).abs().max().item())))
        for bi, x in enumerate(batches):
            rec = refs[bi][2]
            xin = x[:, :, 0] if is_image else x
            print(json.dumps(dict(event="checker", checker=chk.kind, batch=bi, psnr_recon_vs_input=round(psnr(rec, xin), 2),
                                  codes_used=int(refs[bi][0].unique().numel()))))
        return 0
    from omnitokenizer_amd import _lib
    model = model.cuda().eval()
    model.update_codebook_usage_on_encode = False  # a measurement tool: leave the module's state alone
    for name in a.modes.split(","):
        opts = MODES[name]
        tot = dict(tokens=0, flips=0, not_near_tie=0, z_err=0.0, pixel_err=0.0, pixel_rel=0.0, psnr_vs_checker=1e9, psnr_vs_input=[])
        try:
            for k, v in opts.items():
                _lib.set_option(k, v)
            for bi, x in enumerate(batches):
                ids_r, z_r, rec_r = refs[bi]
                xin = (x[:, :, 0] if is_image else x).cuda()
                ids, z = model.encode(xin, is_image, return_latents=True)
                rec_on_ref = model.decode(ids_r.cuda(), is_image).cpu()   # tier (ii): same ids in, pixels out
                rec_own = model.decode(ids, is_image).cpu()
                flips, gaps = near_tie_audit(ids, ids_r, z, E)
                scale = max(1.0, rec_r.abs().max().item())
                perr = (rec_on_ref - rec_r).abs().max().item()
                line = dict(event="batch", mode=name, batch=bi, tokens=ids.numel(), flips=flips, gaps=[float(f"{g:.3e}") for g in gaps],
                            z_err=(z.cpu() - z_r).abs().max().item(), pixel_err=perr, ref_absmax=scale,
                            psnr_vs_checker=round(psnr(rec_on_ref, rec_r), 2),
                            psnr_recon_vs_input=round(psnr(rec_own, xin.cpu()), 2),
                            psnr_checker_recon_vs_input=round(psnr(rec_r, xin.cpu()), 2))
                print(json.dumps(line))
                tot["tokens"] += ids.numel()
                tot["flips"] += flips
                tot["not_near_tie"] += sum(g >= 1e-5 for g in gaps)
                tot["z_err"] = max(tot["z_err"], line["z_err"])
                tot["pixel_err"] = max(tot["pixel_err"], perr)
                tot["pixel_rel"] = max(tot["pixel_rel"], perr / scale)
                tot["psnr_vs_checker"] = min(tot["psnr_vs_checker"], line["psnr_vs_checker"])
                tot["psnr_vs_input"].append(line["psnr_recon_vs_input"])
        finally:
            _lib.set_option("gemm_mode", 2)
            _lib.set_option("attn_mode", 1)
        tot["psnr_vs_input"] = round(float(np.mean(tot["psnr_vs_input"])), 2)
        tot["pass"] = tot["not_near_tie"] == 0 and tot["pixel_rel"] <= a.pixel_tol
        ok = ok and tot["pass"]
        results[name] = tot
        print(json.dumps(dict(event="summary", mode=name, checker=chk.kind, **tot)))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
