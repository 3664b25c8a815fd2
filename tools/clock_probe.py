#!/usr/bin/env python
"""Shader clock and board power while the h2 GEMM runs on random / constant / zero operands (the same binary), and
while one whole encode+decode step runs: the direct reading behind DESIGN.md's "power-limited" statement.
Samples the amdgpu hwmon files (freq1_input = sclk in Hz, power1_average / power1_input in uW) from a host thread;
falls back to `rocm-smi --showclocks --showpower` once per phase when hwmon is not readable.
python tools/clock_probe.py [--seconds 2.0]"""
import argparse
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from omnitokenizer_amd import ops  # noqa: E402
from tools import gpu_power  # noqa: E402


def hwmon_files(load_fn):
    """The hwmon files of the GPU this process computes on (see tools/gpu_power.py)."""
    torch.cuda.synchronize()
    time.sleep(0.5)
    idle = gpu_power.snapshot()
    load_fn()              # queues about a second of work
    time.sleep(0.6)
    busy = gpu_power.snapshot()
    torch.cuda.synchronize()
    return gpu_power.pick_hwmon(idle, busy, pci=gpu_power.pci_address(torch.cuda.current_device()))


def smi_once():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20)
        return " | ".join(l.strip() for l in r.stdout.splitlines() if "sclk" in l or "Power" in l)
    except Exception as e:  # noqa: BLE001
        return f"rocm-smi failed: {e}"


def phase(name, fn, seconds, files, flops=None):
    fn()
    torch.cuda.synchronize()
    s = gpu_power.Sampler(files) if files else None
    smi = {}
    if s:
        s.start()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    per = max((time.time() - t0) / 10, 1e-5)
    n = 0
    ev0.record()
    if files:
        t0 = time.time()
        while time.time() - t0 < seconds:   # batches of ~20 ms keep the queue short and the GPU busy
            for _ in range(max(int(0.02 / per), 1)):
                fn()
                n += 1
            torch.cuda.synchronize()
    else:
        n = max(int(seconds / per), 10)     # one queue of `seconds` of work, read rocm-smi while it drains
        for _ in range(n):
            fn()
        smi["mid"] = smi_once()
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / n
    line = f"{name:28s} {ms:8.3f} ms/launch"
    if flops:
        line += f" {flops / ms / 1e9:6.0f} TF"
    if s:
        s.stop()
        sm = s.summary()
        for k, unit in (("sclk_mhz", "MHz"), ("power_w", "W")):
            if k in sm:
                line += f"  {k.split('_')[0]} avg {sm[k]['avg']:7.1f} min {sm[k]['min']:7.1f} max {sm[k]['max']:7.1f} {unit}"
        line += f"  ({sm['samples']} samples)"
    else:
        line += "  " + smi.get("mid", "")
    print(line, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.0)
    a = ap.parse_args()
    L, D = 32 * 5120, 512
    g = torch.Generator(device="cuda").manual_seed(0)
    flops = 2.0 * L * D * 2730
    x0 = torch.randn(L, D, device="cuda", generator=g)
    pk0 = ops.h2_pack_weight(ops.pack_geglu_weight(torch.randn(2730, D, device="cuda", generator=g) * 0.04, 1408))
    def burst():
        for _ in range(900):
            ops.linear_h2(x0, pk0, 8.0, geglu=True)
    files, info = hwmon_files(burst)
    print("hwmon:", files if files else "not readable; using rocm-smi once per phase", info)
    for data in ("randn", "const", "zeros", "randn"):
        if data == "randn":
            x, w = torch.randn(L, D, device="cuda", generator=g), torch.randn(2730, D, device="cuda", generator=g) * 0.04
        elif data == "const":
            x, w = torch.full((L, D), 0.5, device="cuda"), torch.full((2730, D), 0.5, device="cuda")
        else:
            x, w = torch.zeros(L, D, device="cuda"), torch.zeros(2730, D, device="cuda")
        pk = ops.h2_pack_weight(ops.pack_geglu_weight(w, 1408))
        phase(f"gemm_h2 FF-in, {data} operands", lambda: ops.linear_h2(x, pk, 8.0, geglu=True), a.seconds, files, flops)
    # an HBM-bound kernel for comparison (clock under a memory-bound load)
    y = torch.randn(L, D, device="cuda", generator=g)
    gam = torch.ones(D, device="cuda")
    phase("layernorm (HBM-bound)", lambda: ops.layernorm(y, gam, gam), a.seconds, files)
    # the whole encode + decode step of bench.py's workload
    from omnitokenizer_amd import OmniTokConfig, OmniTokenizer_VQGAN, make_args, synth
    args = make_args(2, resolution=256)
    model = OmniTokenizer_VQGAN(args)
    model.load_state_dict(synth.synth_state_dict(OmniTokConfig.from_args(args), seed=0), strict=True)
    model = model.cuda().eval()
    clips = synth.synth_video(32, 17, 256, seed=1234).cuda().contiguous()

    def step():
        model.decode(model.encode(clips, False), False)
    phase("encode + decode step (C3)", step, max(a.seconds, 3.0), files)


if __name__ == "__main__":
    main()
