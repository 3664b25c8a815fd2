#!/usr/bin/env python
"""bench.py -- patches/sec of OmniTokenizer encode+decode on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

A step = one pass of the hot path over one batch already resident in HBM:
    ids = encode(x_local) ; [N > 1: RCCL all-gather of the ids] ; pixels = decode(ids_local)
Workload (config.workload): BASELINE config C3 -- B = 32 clips of 17x256x256 per GPU, stage-2
(imagenet_k600) architecture, synthetic pixels + seeded random weights (no checkpoints offline).
Weak scaling: every rank processes its own 32 clips (C4 = 256 clips on 8 GPUs); value = all
ranks' patches / max-over-ranks time.  One patch = one latent token = one element of encode()'s id
tensor (5120 per clip).  dtype f32: the path computes in fp32 like the reference.

Extra objects on the JSON line (rank 0):
  roofline      dominant kernel family, from HIP events recorded around every launch on the launch
                stream in a separate profiled pass of the same step
  kernels       per-family ms / achieved rate of that pass (spatial attention, VQ, ...)
  cpu_baseline  the CPU oracle (a port of the reference's arithmetic on ATen/MKL) timed on this
                host on a bounded sample (N = 1 only)
  parity        the TIMED batch against the oracle: `flow` (data flow the timed step ran), id flips over all B clips
                (`flips_all_clips`), latent / pixel error and PSNR of clip 0 taken out of full-batch calls
  also          (default C3 line only) strict-fp32 arithmetic with its own `flips_all_clips`, C2, 8 clips, C5 shape, one-image /
                one-clip latency, the heavy-statistics reference fixture per arithmetic / data-flow arm
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 MFMA = fp32 vector peak
PEAK_F16_TFLOPS = 2500.0  # dense fp16 / bf16 MFMA peak (same guide)
PEAK_HBM_GBS = 8000.0
# matrix-pipe products a GEMM kernel spends per fp32 multiply-add, by engine "gemm_mode"
GEMM_MODES = {0: ("fp32-input MFMA", 1, PEAK_F32_TFLOPS),
              1: ("fp32 operands as 3 bf16 planes, 6 bf16-MFMA products", 6, PEAK_F16_TFLOPS),
              2: ("fp32 operands as 2 fp16 planes, 3 fp16-MFMA products", 3, PEAK_F16_TFLOPS)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU (C3: 32)")
    ap.add_argument("--frames", type=int, default=17)
    ap.add_argument("--resolution", type=int, default=256)
    ap.add_argument("--n-codes", type=int, default=0, help="codebook size (default: the stage-2 8192; C5: 16384)")
    ap.add_argument("--stage", type=int, default=2, choices=[1, 2],
                    help="architecture of the released configurations: 2 = imagenet_k600 (rope, pt = 4; the headline), 1 = imagenet_only "
                         "(relative-position bias + legacy attention, pt = 2; BASELINE config 1's architecture, images)")
    ap.add_argument("--gemm-mode", type=int, default=2, choices=[0, 1, 2],
                    help="engine GEMM arithmetic: 2 fp16x2 split (default), 1 bf16x3 split, 0 fp32-input MFMA")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=INT",
                    help="omnitok_set_option switch for A/B measurements (e.g. attn_vpack=0); recorded in config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clock-probe", action="store_true", help="skip the hwmon clock / power sampling (profiled runs)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--native-gather", action="store_true",
                    help="N > 1: issue the id all-gather from C++ (ncclAllGather through libomnitok.so, include/omnitok_comm.h) "
                         "instead of torch.distributed; without the flag the native path is still probed after the timed region")
    ap.add_argument("--no-also", action="store_true", help="skip the C2 / C5 / strict-fp32 / heavy-parity extras of the default C3 line")
    a = ap.parse_args()

    from omnitokenizer_amd import launch

    if a.gpus > 1 and "LOCAL_RANK" not in os.environ and torch.cuda.device_count() < a.gpus:
        sys.exit(f"bench.py --gpus {a.gpus}: this node exposes {torch.cuda.device_count()} GPU(s) "
                 "(one rank per GPU; no oversubscription)")
    # `python bench.py --gpus N` with no launcher around it: start the N ranks ourselves
    rc = launch.maybe_respawn(os.path.abspath(__file__), sys.argv[1:], a.gpus)
    if rc is not None:
        sys.exit(rc)
    with launch.rank_errors():   # a failing rank names itself and its reason before the launcher tears the job down
        run(a)


def run(a):
    from omnitokenizer_amd import launch
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU path)"
    info = launch.init_ranks(a.gpus, backend="nccl")   # RCCL on ROCm
    world, rank, local_rank = info.world, info.rank, info.local_rank

    from omnitokenizer_amd import OmniTokenizer_VQGAN, _lib, make_args, synth
    from omnitokenizer_amd.config import OmniTokConfig

    _lib.set_option("gemm_mode", a.gemm_mode)
    for kv in a.option:
        name, _, val = kv.partition("=")
        _lib.set_option(name, int(val))
    over = dict(resolution=a.resolution)
    if a.n_codes:
        over["n_codes"] = a.n_codes
    if a.frames > 17:
        over["sequence_length"] = a.frames
    args = make_args(a.stage, **over)
    att_mode = "legacy" if a.stage == 1 else "sdpa"   # README.md:58: imagenet_only.ckpt runs the legacy attention
    cfg = OmniTokConfig.from_args(args, attention_mode=att_mode)
    sd = synth.synth_state_dict(cfg, seed=0)
    model = OmniTokenizer_VQGAN(args, attention_mode=att_mode)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()

    B = a.batch
    is_image = a.frames == 1
    # every rank generates its own shard (distinct seed per rank, B DISTINCT clips): the batch is
    # sharded by clip
    x = (synth.synth_image(B, a.resolution, seed=1234 + rank) if is_image
         else synth.synth_video(B, a.frames, a.resolution, seed=1234 + rank)).cuda().contiguous()
    n_total = B * world

    if world > 1:  # one data flow for every rank, decided from the global batch (ragged shards would otherwise differ)
        from omnitokenizer_amd import dist as od
        t_, h_, w_ = model.latent_dims(a.frames, a.resolution, a.resolution)
        od.pin_data_flow(model, n_total, t_ * h_ * w_)
    res = launch.timed_sharded_steps(info, lambda xs: model.encode(xs, is_image),
                                     lambda i: model.decode(i, is_image), x, a.steps, a.warmup,
                                     native_gather=a.native_gather)
    dt, ids = res.seconds, res.ids_local
    # N > 1: the C++ ncclAllGather path (omnitok_comm.h) checked against the gathered ids of the timed region and timed,
    # outside the timed region and never fatal -- so that the first multi-GPU run exercises it whichever path was timed
    native_probe = launch.probe_native_gather(info, ids, res) if world > 1 else None
    tokens_per_clip = ids.numel() // B
    value = n_total * tokens_per_clip * a.steps / dt

    out = None
    if rank == 0:
        # ---- profiled pass: HIP events around every launch, on the launch stream ---------------
        model.set_timing(True)
        model.timing_report()
        nprof = 2
        for _ in range(nprof):
            model.encode(x, is_image)
            model.decode(ids, is_image)
        rep = model.timing_report()
        model.set_timing(False)
        kernels = {}
        for name, r in rep.items():
            ms = r["ms"] / nprof
            kernels[name] = dict(ms_per_step=round(ms, 4), launches_per_step=r["calls"] // nprof,
                                 work_per_step=r["work"] / nprof)
        mfma = {"gemm_ff_in", "gemm_ff_out", "gemm_qkv", "gemm_out", "gemm_patch", "gemm_pixels", "attn_spatial",
                "attn_window", "vq_argmin"}
        gm_name, gm_products, gm_pipe_peak = GEMM_MODES[a.gemm_mode]
        gemm_roof = gm_pipe_peak / gm_products   # fp32-equivalent roof of the GEMM kernels in this mode
        # the roof each MFMA kernel family is graded against: the split-operand kernels (GEMMs in gemm_mode 1 / 2, spatial
        # attention in gemm_mode 2 with attn_mode 1) run on the fp16 / bf16 pipe with `products` MFMAs per fp32 multiply-add;
        # window attention and the quantiser stay on the fp32-input MFMA
        opts = dict(kv.split("=") for kv in a.option)
        attn_split = a.gemm_mode == 2 and int(opts.get("attn_mode", 1)) == 1
        roofs = {n: gemm_roof for n in mfma if n.startswith("gemm_")}
        roofs["attn_spatial"] = PEAK_F16_TFLOPS / 3 if attn_split else PEAK_F32_TFLOPS
        # window attention runs on the split-operand pipe too since r04 ("attn_window_mode" 1); it is priced against HBM as well
        # below (16 flop/B -- its roof is bandwidth: packed q, k, v in + planes out = 4 L D 4 bytes per launch)
        win_split = a.gemm_mode == 2 and int(opts.get("attn_window_mode", 1)) == 1 and int(opts.get("gemm_pl", 1)) == 1
        roofs["attn_window"] = PEAK_F16_TFLOPS / 3 if win_split else PEAK_F32_TFLOPS
        roofs["vq_argmin"] = PEAK_F32_TFLOPS
        for name, k in kernels.items():
            if k["ms_per_step"] > 0:
                if name in mfma:
                    # algorithmic (fp32-equivalent) rate against the roof of the arithmetic the kernel runs on
                    k["tflops"] = round(k["work_per_step"] / (k["ms_per_step"] * 1e-3) / 1e12, 2)
                    k["roof_tflops"] = round(roofs[name], 1)
                    k["frac_of_mode_roof"] = round(k["tflops"] / roofs[name], 4)
                else:
                    k["gbs"] = round(k["work_per_step"] / (k["ms_per_step"] * 1e-3) / 1e9, 1)
                    k["frac_hbm_peak"] = round(k["gbs"] / PEAK_HBM_GBS, 4)
        dom = max(kernels, key=lambda n: kernels[n]["ms_per_step"])
        dk = kernels[dom]
        is_gemm = dom.startswith("gemm_")
        if dom in mfma:
            per_launch_flops = dk["work_per_step"] / dk["launches_per_step"]
            avg_ms = dk["ms_per_step"] / dk["launches_per_step"]
            peak = roofs[dom]
            roofline = dict(kernel=dom, bound="mfma", achieved=round(per_launch_flops / (avg_ms * 1e-3) / 1e12, 2),
                            peak=round(peak, 1), unit="TFLOP/s", avg_launch_ms=round(avg_ms, 4),
                            flops_per_launch=per_launch_flops, traffic=None)
            if is_gemm:
                roofline["peak_definition"] = (f"{gm_pipe_peak:.0f} TF dense MFMA peak / {gm_products} matrix-pipe "
                                               f"products per fp32 multiply-add ({gm_name}); algorithmic fp32 flops")
                roofline["matrix_pipe_tflops"] = round(roofline["achieved"] * gm_products, 1)
        else:
            per_launch_bytes = dk["work_per_step"] / dk["launches_per_step"]
            avg_ms = dk["ms_per_step"] / dk["launches_per_step"]
            roofline = dict(kernel=dom, bound="hbm", achieved=round(per_launch_bytes / (avg_ms * 1e-3) / 1e9, 1),
                            peak=PEAK_HBM_GBS, unit="GB/s", avg_launch_ms=round(avg_ms, 4),
                            bytes_per_launch=per_launch_bytes, traffic=None)
        roofline["frac"] = round(roofline["achieved"] / roofline["peak"], 4)
        # Counter-derived fields (HBM-side traffic per launch, matrix-pipe busy cycles): bench.py cannot collect PMC
        # counters itself, they come from profiles/pmc_traffic.json -- rocprofv3 --pmc passes of this very command
        # (tools/pmc_collect.sh), stamped with the sha256 of the csrc/ tree they were measured on.  A stamp that does not
        # match the sources of the library in use nulls them: a driver-run line never carries counters of other code.
        pmc, pmc_state = {}, "profiles/pmc_traffic.json missing"
        try:
            from tools.pmc_to_json import csrc_digest
            pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            stamp = pj.get("stamp", {})
            if stamp.get("csrc_sha256") != csrc_digest(ROOT):
                pmc_state = (f"stale: profiles/pmc_traffic.json was measured on csrc {str(stamp.get('csrc_sha256'))[:12]} "
                             f"(HEAD {str(stamp.get('git_head'))[:10]}), this tree is {csrc_digest(ROOT)[:12]}")
            elif not (B == 32 and a.frames == 17 and a.resolution == 256 and cfg.n_codes == 8192 and a.gemm_mode == 2
                      and a.stage == 2 and not a.option):
                pmc_state = "the committed PMC passes cover the default C3 command only"
            else:
                pmc = pj.get(f"gemm_mode_{a.gemm_mode}", {})
                pmc_state = f"ok: stamp {stamp.get('csrc_sha256', '')[:12]} (HEAD {str(stamp.get('git_head'))[:10]}, {stamp.get('tag')})"
            if is_gemm and "power_limit_evidence" in pj:
                roofline["power_limit_evidence"] = pj["power_limit_evidence"]
        except Exception as ex:  # noqa: BLE001
            pmc_state = f"unreadable: {ex!r}"
        roofline["pmc_fields"] = pmc_state
        if dom in pmc:
            if "read_bytes" in pmc[dom]:
                roofline["traffic"] = pmc[dom]["read_bytes"] + pmc[dom]["write_bytes"]
                roofline["traffic_source"] = pmc[dom]["source"]
            if "mfma_busy_pct" in pmc[dom]:
                roofline["mfma_busy_pct_pmc"] = pmc[dom]["mfma_busy_pct"]
                roofline["mfma_busy_source"] = pmc[dom]["mfma_source"]
        # algorithmic bytes of the dominant GEMM families at this workload: A operand + output (+ residual) + weight
        Ltok = B * tokens_per_clip
        alg_bytes = {"gemm_ff_in": Ltok * (512 + 1408) * 4 + 2816 * 512 * 4,
                     "gemm_ff_out": Ltok * (1408 + 512 + 512) * 4 + 512 * 1408 * 4,
                     "gemm_qkv": Ltok * (512 + 1536) * 4 + 1536 * 512 * 4}.get(dom)
        if alg_bytes:
            roofline["algorithmic_bytes"] = alg_bytes
        # the two kernels north_star names, with its formulas: always on the line; the counter-derived
        # parts are null when the PMC stamp does not match
        for name in ("attn_spatial", "vq_argmin"):
            if name in kernels and kernels[name]["ms_per_step"] > 0:
                k, e = kernels[name], pmc.get(name, {})
                k["mfma_busy_pct"] = e.get("mfma_busy_pct")
                tr = e["read_bytes"] + e["write_bytes"] if "read_bytes" in e else None
                k["traffic_bytes_per_launch"] = tr
                k["traffic_gbs"] = round(tr * k["launches_per_step"] / (k["ms_per_step"] * 1e-3) / 1e9, 1) if tr else None
        if "attn_window" in kernels and kernels["attn_window"]["ms_per_step"] > 0:
            kw = kernels["attn_window"]
            kw["algorithmic_gbs"] = round(kw["launches_per_step"] * 4.0 * Ltok * 512 * 4 / (kw["ms_per_step"] * 1e-3) / 1e9, 1)
            kw["frac_hbm_peak"] = round(kw["algorithmic_gbs"] / PEAK_HBM_GBS, 4)
            if win_split:   # 16 flop/B on the packed-operand path: bandwidth is the roof that binds, the matrix-pipe fraction says nothing
                kw["bound"] = "hbm"
                kw.pop("frac_of_mode_roof", None)
                kw.pop("roof_tflops", None)
        if "vq_argmin" in kernels and kernels["vq_argmin"]["ms_per_step"] > 0:
            vq_bytes = Ltok * 40 + cfg.n_codes * 32
            kernels["vq_argmin"]["compulsory_gbs"] = round(vq_bytes / (kernels["vq_argmin"]["ms_per_step"] * 1e-3) / 1e9, 2)
            # frac_f32_peak = algorithmic distance flops (2 N n_codes 8) / family time / the fp32-MFMA peak: the roof
            # of the EXACT sweep.  Since r05 the default search screens with one fp16 MFMA per 32 x 32 tile and re-evaluates only
            # the candidates exactly ("vq_screen" 1, same ids bit for bit): it does a fifth of that matrix work, is VALU-bound
            # (the min tree), and this fraction can exceed what the fp32 pipe could deliver
            kernels["vq_argmin"]["frac_f32_peak"] = kernels["vq_argmin"].get("frac_of_mode_roof")
            kernels["vq_argmin"]["search"] = ("screened: fp16-MFMA coarse distance with a rigorous error bound + exact fp32 "
                                              "re-evaluation of the candidate tiles" if int(opts.get("vq_screen", 1)) and
                                              not cfg.use_external_codebook else "exact fp32-MFMA sweep")

        if is_image:
            wl_name = "C2" if (B, a.resolution) == (64, 256) else "images"
        elif (a.frames, a.resolution) == (17, 256):
            wl_name = "C3" if world == 1 else f"C4-shape ({n_total} clips over {world} GPUs)"
        elif (a.frames, a.resolution, cfg.n_codes) == (65, 512, 16384):
            wl_name = "C5"
        else:
            wl_name = "clips"
        out = {
            "metric": "patches/sec encode+decode", "value": round(value, 1), "unit": "patches/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if a.gemm_mode == 0 else (f"f32 ({gm_name}, fp32 accumulate; "
                                                           + ("spatial attention likewise; " if a.gemm_mode == 2 else "attention and ")
                                                           + "VQ: fp16-MFMA screen + exact fp32 re-evaluation, ids bit-exact)"),
            "gemm_mode": a.gemm_mode, "data": "synthetic",
            "config": {"workload": wl_name + f": B={B}/GPU " + (f"{a.resolution}x{a.resolution} images" if is_image
                                    else f"{a.frames}x{a.resolution}x{a.resolution} clips")
                                   + (f", stage-2 (imagenet_k600 arch: ttww/tttt, pt=4, rope, n_codes={cfg.n_codes}), encode + " if a.stage == 2 else
                                      f", stage-1 (imagenet_only arch: ttww/tttt, pt=2, relative-position bias, legacy attention, n_codes={cfg.n_codes}), encode + ")
                                   + f"{'RCCL id all-gather + ' if world > 1 else ''}decode; {B} distinct clips per rank",
                       "global_batch": n_total, "tokens_per_clip": tokens_per_clip,
                       "parallelism": f"clip-sharded x{world}", **({"options": a.option} if a.option else {})},
            "rccl_world_size": res.world_seen, "ids_crc32": res.ids_crc, "allgather_ms": res.allgather_ms,
            "gather_impl": res.gather_impl, "native_gather_probe": native_probe,
            "per_rank_ms": {"min": min(res.per_rank_ms), "max": max(res.per_rank_ms), "all": res.per_rank_ms},
            "roofline": roofline, "launches_per_step": sum(k["launches_per_step"] for k in kernels.values()),
            "kernels": kernels,
            "workspace_gb": round(model.workspace_bytes() / 2**30, 2),
        }

        # ---- shader clock / board power during the same step (host thread reading the amdgpu hwmon files) ------
        # The dominant kernels run at the board's power cap with the clock well under 2400 MHz on real data
        # (profiles/r02_clock_power_probe.txt); this puts the live reading next to the roofline fractions.
        out["clock_power"] = None
        if world == 1 and not a.no_clock_probe:
            try:
                from tools import gpu_power
                torch.cuda.synchronize()
                time.sleep(0.5)
                idle = gpu_power.snapshot()
                n_pw = max(2, int(1.5 / max(dt / a.steps, 1e-3)))
                snaps, watching = [], [True]

                def watch():   # decode() synchronises (id check), so the load is only visible from another thread
                    while watching[0]:
                        snaps.append(gpu_power.snapshot())
                        time.sleep(0.05)
                import threading
                th = threading.Thread(target=watch, daemon=True)
                th.start()
                for _ in range(n_pw):
                    model.decode(model.encode(x, is_image), is_image)
                torch.cuda.synchronize()
                watching[0] = False
                th.join()
                busy = {d: max(sn.get(d, 0) for sn in snaps) for d in idle}
                files, hw = gpu_power.pick_hwmon(idle, busy, pci=gpu_power.pci_address(torch.cuda.current_device()))
                if files:
                    smp = gpu_power.Sampler(files)
                    smp.start()
                    for _ in range(n_pw):
                        model.decode(model.encode(x, is_image), is_image)
                        torch.cuda.synchronize()
                    smp.stop()
                    cp = smp.summary(drop_first=0.1)
                    cp.update(hw)
                    cp["note"] = ("amdgpu hwmon freq1_input / power1_input sampled every 10 ms over %d more steps; the fp16 "
                                  "MFMA peak scales with sclk / 2400 MHz" % n_pw)
                    if "sclk_mhz" in cp and roofline.get("bound") == "mfma":
                        roofline["frac_at_step_avg_clock"] = round(roofline["frac"] * 2400.0 / cp["sclk_mhz"]["avg"], 4)
                    out["clock_power"] = cp
            except Exception as e:  # noqa: BLE001  (measurement extra: never fails the bench)
                out["clock_power"] = {"error": repr(e)}

        # ---- CPU baseline + parity on a bounded sample ---------------------------------------------
        if world == 1 and not a.no_cpu_baseline:
            from oracle import omnitok_oracle as orc
            ncpu = os.cpu_count() or 1
            xs = x[:1].cpu()
            # pick the thread count that serves this host best (MKL on a many-core box is slower
            # with every SMT thread than with a subset), on a quick image-sized probe
            probe = xs if is_image else xs[:, :, 0].contiguous()
            best_thr, best_t = 1, float("inf")
            with torch.no_grad():
                for thr in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128)}):
                    torch.set_num_threads(thr)
                    orc.encode(sd, probe, True, cfg)
                    t = time.perf_counter()
                    orc.encode(sd, probe, True, cfg)
                    t = time.perf_counter() - t
                    if t < best_t:
                        best_thr, best_t = thr, t
            ncpu_used = best_thr
            torch.set_num_threads(ncpu_used)
            # what is timed: the REFERENCE ITSELF wherever it is mounted (the build container: /root/reference through the
            # stub harness of SURVEY.md 8(c), kind "reference"); on the GPU box, which has no /root/reference, the oracle
            # port (kind "port", calibrated against the reference below)
            from oracle import ref_harness as rh
            ref_model = None
            if rh.reference_available():
                try:
                    ref_model = rh.build_reference_model(args)
                    ref_model.load_state_dict(sd, strict=False)
                except Exception:  # noqa: BLE001
                    ref_model = None
            xin = xs[:, :, 0].contiguous() if (is_image and xs.dim() == 5) else xs

            def cpu_step():
                if ref_model is not None:
                    with rh.attention_mode(cfg.attention_mode):
                        return ref_model.decode(ref_model.encode(xin, is_image), is_image)
                return orc.decode(sd, orc.encode(sd, xs, is_image, cfg), is_image, cfg)
            with torch.no_grad():
                taps = {}
                ids_ref = orc.encode(sd, xs, is_image, cfg, taps=taps)   # warm-up, also the parity reference
                rec_ref = orc.decode(sd, ids_ref, is_image, cfg)
                if ref_model is not None:
                    cpu_step()
                reps, spent = 0, 0.0
                while reps < 1 or (spent < a.cpu_seconds and reps < 50):
                    t = time.perf_counter()
                    cpu_step()
                    spent += time.perf_counter() - t
                    reps += 1
            per = spent / reps
            try:
                cpu_model = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
            except Exception:
                cpu_model = "unknown"
            try:   # the port calibrated against the reference itself (build container, same clip / threads)
                cal = json.load(open(os.path.join(ROOT, "profiles", "cpu_reference_vs_port.json")))
                ref_over_port = cal["reference_over_port"]
                cal_src = (f"profiles/cpu_reference_vs_port.json: reference {cal['reference_patches_per_s']} vs port "
                           f"{cal['port_patches_per_s']} patches/s on {cal['threads']} threads of {cal['cpu']}, "
                           f"ids equal, pixel diff {cal['pixel_max_abs_diff']}")
            except Exception:
                ref_over_port, cal_src = None, None
            out["cpu_baseline"] = {"value": round(tokens_per_clip / per, 1), "unit": "patches/s", "cores": ncpu_used,
                                   "kind": "reference" if ref_model is not None else "port",
                                   "reference_over_port": ref_over_port,
                                   "reference_over_port_source": cal_src,
                                   "host_logical_cpus": ncpu, "cpu": cpu_model,
                                   "sample": f"1 clip {a.frames}x{a.resolution}x{a.resolution} encode+decode, "
                                             f"{reps} timed reps after 1 warm-up, "
                                             + ("the unmodified reference (oracle/ref_harness.py)" if ref_model is not None
                                                else "torch CPU fp32 oracle port")
                                             + f" (ATen/MKL, {ncpu_used} threads = best of a 8..128 probe)"}
            # Parity of the TIMED work: the ids are the timed batch's own (last timed step, all B clips in one call = the data
            # flow and tile schedule that was timed), the latents / pixels of clip 0 come out of a full-batch encode / decode.
            # The oracle encodes every clip of the batch (a few clips per call to bound host memory): flips_all_clips.
            with torch.no_grad():
                ids_ref_all = [ids_ref]
                t_or = time.perf_counter()
                for c0 in range(1, B, 4):
                    ids_ref_all.append(orc.encode(sd, x[c0:c0 + 4].cpu(), is_image, cfg))
                ids_ref_all = torch.cat(ids_ref_all, 0)
                t_or = time.perf_counter() - t_or
            g_ids_all, g_z = model.encode(x, is_image, return_latents=True)
            assert torch.equal(g_ids_all, ids), "encode() of the timed batch is not deterministic"
            g_rec = model.decode(ids_ref_all.cuda(), is_image)[:1].cpu()
            flips_by_clip = (g_ids_all.cpu() != ids_ref_all).flatten(1).sum(1)
            L_call = int(ids.numel())
            pl_min = _lib.get_option("pl_min_tokens")
            out["parity"] = {"flow": "planes" if (_lib.get_option("gemm_pl") and L_call >= pl_min and a.gemm_mode == 2) else
                                     ("fp32 activations" if a.gemm_mode == 2 else GEMM_MODES[a.gemm_mode][0]),
                             "tokens_per_call": L_call,
                             "source": "ids of the timed batch (last timed step); z / pixels of clip 0 from full-batch encode / decode",
                             "flips_all_clips": int(flips_by_clip.sum()), "ids_all_clips": int(ids_ref_all.numel()),
                             "clips_with_flips": int((flips_by_clip > 0).sum()),
                             "id_flips_vs_oracle": int(flips_by_clip[0]), "ids": int(ids_ref.numel()),
                             "z_max_abs_err": float((g_z[:1].cpu().reshape(taps["z"].shape) - taps["z"]).abs().max()),
                             "pixel_max_abs_err": float((g_rec - rec_ref).abs().max()),
                             "psnr_vs_ref_db": round(orc.psnr(g_rec, rec_ref), 2),
                             "psnr_vs_input_db": round(orc.psnr(g_rec, xs), 2),
                             "oracle_encode_all_clips_s": round(t_or, 1)}
            parity_ref_ids = ids_ref_all
        # ---- driver-observed extras of the default C3 line ------------------------------
        if world == 1 and wl_name == "C3" and B == 32 and a.gemm_mode == 2 and a.stage == 2 and not a.option and not a.no_also:
            out["also"] = also_extras(model, x, sd, a, locals().get("parity_ref_ids"))
        print(json.dumps(out), flush=True)
    launch.finish(info, native_timed=bool(a.native_gather))


def _time_steps(model, x, is_image, steps, warmup=1):
    for _ in range(warmup):
        ids = model.encode(x, is_image)
        model.decode(ids, is_image)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        ids = model.encode(x, is_image)
        model.decode(ids, is_image)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3, ids


def also_extras(model, x, sd, a, ids_ref_all=None):
    """Short extra measurements put on the default line so that they are driver-observed, each a few steps and none inside
    the timed region: the other two single-GPU BASELINE configurations (C2 images, C5-shape long clip), the same C3 step
    in strict fp32 arithmetic (fp32-input MFMA GEMMs and attention: `gemm_mode` 0 / `attn_mode` 0), and the default
    arithmetic against a heavy-statistics reference fixture (tests/golden/heavy_*: outputs of the reference itself on
    trained-like weights, with the reference's own fp32-vs-fp64 noise as the yardstick)."""
    from omnitokenizer_amd import OmniTokenizer_VQGAN, make_args, synth
    from omnitokenizer_amd.config import OmniTokConfig
    also = {}
    try:
        # C3 in strict fp32: per-engine options, restored afterwards
        model.set_option("gemm_mode", 0)
        model.set_option("attn_mode", 0)
        ms, ids_strict = _time_steps(model, x, False, steps=2, warmup=1)
        model.set_option("gemm_mode", -1)
        model.set_option("attn_mode", -1)
        also["strict_fp32"] = {"gemm_mode": 0, "attn_mode": 0, "ms_per_step": round(ms, 3),
                               "patches_s": round(x.shape[0] * 5120 / ms * 1e3, 1),
                               "flips_all_clips": (int((ids_strict.cpu() != ids_ref_all).sum()) if ids_ref_all is not None else None),
                               "ids_all_clips": int(ids_strict.numel()),
                               "note": "same C3 batch, fp32-input MFMA GEMMs and attention (the reference's arithmetic class); 2 steps"}
    except Exception as e:  # noqa: BLE001
        also["strict_fp32"] = {"error": repr(e)}
    try:
        xi = synth.synth_image(64, 256, seed=1234).cuda().contiguous()
        ms, ids = _time_steps(model, xi, True, steps=3)
        also["c2"] = {"workload": "C2: B=64 256x256 images", "ms_per_step": round(ms, 3),
                      "patches_s": round(ids.numel() / ms * 1e3, 1), "steps": 3}
        del xi
    except Exception as e:  # noqa: BLE001
        also["c2"] = {"error": repr(e)}
    try:   # mid-size call: 8 of the timed clips in one call (tail schedule of the plane GEMM)
        ms, ids = _time_steps(model, x[:8].contiguous(), False, steps=5)
        also["clips8"] = {"workload": "8 clips 17x256x256 (the first 8 of the timed batch)", "ms_per_step": round(ms, 3),
                          "patches_s": round(ids.numel() / ms * 1e3, 1), "steps": 5}
    except Exception as e:  # noqa: BLE001
        also["clips8"] = {"error": repr(e)}
    try:
        args5 = make_args(2, resolution=512, n_codes=16384, sequence_length=65)
        cfg5 = OmniTokConfig.from_args(args5)
        m5 = OmniTokenizer_VQGAN(args5)
        m5.load_state_dict(synth.synth_state_dict(cfg5, seed=0), strict=True)
        m5 = m5.cuda().eval()
        x5 = synth.synth_video(1, 65, 512, seed=1234).cuda().contiguous()
        ms, ids = _time_steps(m5, x5, False, steps=3)
        also["c5"] = {"workload": "C5 shape: 1 clip 65x512x512, n_codes 16384", "ms_per_step": round(ms, 3),
                      "patches_s": round(ids.numel() / ms * 1e3, 1), "steps": 3}
        del m5, x5
    except Exception as e:  # noqa: BLE001
        also["c5"] = {"error": repr(e)}
    try:
        # interactive sizes (the callers transformer_eval.py:71,120 / Diffusion/DiT/sample_ddp.py:162 decode a handful of
        # samples): encode + decode latency of ONE image and ONE clip with the same weights, synchronised per call
        lat = {}
        for tag, xs_, img in (("1img", synth.synth_image(1, 256, seed=7).cuda(), True),
                              ("1clip", synth.synth_video(1, 17, 256, seed=7).cuda(), False)):
            for _ in range(3):
                model.decode(model.encode(xs_, img), img)
            torch.cuda.synchronize()
            reps = 20
            t = time.perf_counter()
            for _ in range(reps):
                model.decode(model.encode(xs_, img), img)
                torch.cuda.synchronize()
            lat[f"latency_{tag}_ms"] = round((time.perf_counter() - t) / reps * 1e3, 3)
        also.update(lat)
        also["latency_note"] = "encode+decode of one 256x256 image / one 17x256x256 clip, host-synchronised per call, 20 reps"
    except Exception as e:  # noqa: BLE001
        also["latency_error"] = repr(e)
    try:
        from tests.helpers import GoldenCase
        c = GoldenCase("heavy_s2_sdpa_r256_vid17")
        mh = OmniTokenizer_VQGAN(c.args, attention_mode=c.mode)
        mh.load_state_dict(c.sd, strict=True)
        mh = mh.cuda().eval()
        r = {}
        # the fixture is ONE clip = 5120 tokens: since r06 it runs the plane flow the timed C3 batch runs (one data flow at every
        # size); the fp32-activation flow the "pl_min_tokens" option still selects keeps its own arm
        for tag, gm, am, mt in (("default_plane_flow", -1, -1, -1), ("fp32_activation_flow_option", -1, -1, 1 << 30),
                                ("strict_fp32", 0, 0, -1)):
            mh.set_option("gemm_mode", gm)
            mh.set_option("attn_mode", am)
            mh.set_option("pl_min_tokens", mt)
            ids, z = mh.encode(c.x.cuda(), False, return_latents=True)
            rec = mh.decode(c.ids.cuda(), False)
            zerr = float((z.cpu() - c.z).abs().max())
            perr = float((c.strided(rec.cpu()) - c.recon).abs().max())
            r[tag] = {"flow": "fp32-input MFMA kernels" if gm == 0 else
                              ("fp32 activations (pl_min_tokens option)" if mt > 0 else "planes (what the timed C3 step runs; the default at every size)"),
                      "id_flips_vs_reference": int((ids.cpu() != c.ids).sum()), "ids": int(c.ids.numel()),
                      "z_max_abs_err": zerr, "z_err_over_reference_fp32_noise": round(zerr / c.fp32_noise_z, 2),
                      "pixel_max_abs_err": perr, "pixel_err_over_reference_fp32_noise": round(perr / c.fp32_noise_pix, 2)}
        r["fixture"] = ("tests/golden/heavy_s2_sdpa_r256_vid17.npz: the reference's own outputs on heavy-tailed weights, one "
                        f"17x256x256 clip; its fp32-vs-fp64 noise: z {c.fp32_noise_z:.2e}, pixels {c.fp32_noise_pix:.2e} "
                        f"(|pixel| up to {c.recon_absmax:.0f})")
        also["parity_heavy"] = r
    except Exception as e:  # noqa: BLE001
        also["parity_heavy"] = {"error": repr(e)}
    return also


if __name__ == "__main__":
    main()
