#!/usr/bin/env python
"""Per-kernel-family time of one encode+decode (engine HIP-event timing) for a given batch shape.
    python tools/breakdown.py [--frames 1] [--batch 1] [--resolution 256]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from omnitokenizer_amd import OmniTokenizer_VQGAN, make_args, synth  # noqa: E402
from omnitokenizer_amd.config import OmniTokConfig  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--resolution", type=int, default=256)
    a = ap.parse_args()
    args = make_args(2, resolution=a.resolution)
    cfg = OmniTokConfig.from_args(args)
    m = OmniTokenizer_VQGAN(args)
    m.load_state_dict(synth.synth_state_dict(cfg, 0), strict=True)
    m = m.cuda().eval()
    is_image = a.frames == 1
    x = (synth.synth_image(a.batch, a.resolution) if is_image
         else synth.synth_video(a.batch, a.frames, a.resolution)).cuda()
    for _ in range(3):
        ids = m.encode(x, is_image)
        m.decode(ids, is_image)
    m.set_timing(True)
    m.timing_report()
    n = 5
    for _ in range(n):
        ids = m.encode(x, is_image)
        m.decode(ids, is_image)
    rep = m.timing_report()
    tot = sum(r["ms"] for r in rep.values()) / n
    for k, r in sorted(rep.items(), key=lambda kv: -kv[1]["ms"]):
        print(f"{k:18s} launches {r['calls'] // n:4d}  {r['ms'] / n:8.3f} ms  {100 * r['ms'] / n / tot:5.1f}%  "
              f"{r['ms'] / r['calls'] * 1e3:7.1f} us/launch")
    print(f"{'sum':18s} {tot:8.3f} ms")


if __name__ == "__main__":
    main()
