#!/usr/bin/env python
"""Calibrates bench.py's cpu_baseline (kind "port" = the CPU oracle) against the REFERENCE itself: times
OmniTokenizer_VQGAN.encode()+decode() of the unmodified reference (imported from /root/reference through
oracle/ref_harness.py) and of the oracle on the same 17x256x256 clip, same weights, same thread count, in the
build container (the GPU box has no /root/reference).  Writes profiles/cpu_reference_vs_port.json; bench.py
reports its ratio as cpu_baseline.reference_over_port.

    python tools/cpu_baseline_calibrate.py [--threads 8] [--reps 3]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from omnitokenizer_amd import synth  # noqa: E402
from omnitokenizer_amd.config import OmniTokConfig, make_args  # noqa: E402
from oracle import omnitok_oracle as orc  # noqa: E402
from oracle import ref_harness as rh  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    assert rh.reference_available(), "needs /root/reference (build container)"
    torch.set_num_threads(a.threads)
    args = make_args(2, resolution=256)
    cfg = OmniTokConfig.from_args(args)
    sd = synth.synth_state_dict(cfg, seed=0)
    model = rh.build_reference_model(args)
    model.load_state_dict(sd, strict=False)
    x = synth.synth_video(1, 17, 256, seed=1234)

    def t_ref():
        with torch.no_grad(), rh.attention_mode("sdpa"):
            ids = model.encode(x, False)
            return model.decode(ids, False), ids

    def t_port():
        with torch.no_grad():
            ids = orc.encode(sd, x, False, cfg)
            return orc.decode(sd, ids, False, cfg), ids

    res = {}
    outs = {}
    for name, fn in (("reference", t_ref), ("port", t_port)):
        outs[name] = fn()  # warm-up
        best = float("inf")
        for _ in range(a.reps):
            t = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t)
        res[name] = best
    assert torch.equal(outs["reference"][1], outs["port"][1]), "oracle ids differ from the reference"
    perr = (outs["reference"][0] - outs["port"][0]).abs().max().item()
    try:
        cpu = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        cpu = "unknown"
    out = {"clip": "1 x 17x256x256, stage-2 architecture, encode + decode", "threads": a.threads, "cpu": cpu,
           "reference_s": round(res["reference"], 4), "port_s": round(res["port"], 4),
           "reference_patches_per_s": round(5120 / res["reference"], 1), "port_patches_per_s": round(5120 / res["port"], 1),
           "reference_over_port": round(res["port"] / res["reference"], 4),
           "ids_equal": True, "pixel_max_abs_diff": perr,
           "note": "reference_over_port = reference throughput / port throughput on the same host and threads"}
    path = os.path.join(ROOT, "profiles", "cpu_reference_vs_port.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
