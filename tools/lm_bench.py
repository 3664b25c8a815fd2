#!/usr/bin/env python
"""Tokens/s of the LM consumer's sampling loop (sample_with_past) on the MI355X, reference k600
LM shape by default (scripts/lm_train/train_k600.sh: vocab 8192, block 5120, 24 layers, 16 heads,
1536 wide), synthetic weights.  One JSON line: throughput, per-step time at the start / end of the
sequence, the HBM roofline of a step (weight bytes + K/V bytes read) and a CPU baseline (the oracle's
KV-cached step on the host, bounded sample).
    python tools/lm_bench.py [--batch 1] [--steps 512] [--ctx 0] [--no-cpu-baseline]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from omnitokenizer_amd import gpt as og  # noqa: E402
from omnitokenizer_amd.synth import synth_gpt_state  # noqa: E402

PEAK_HBM_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--ctx", type=int, default=0, help="tokens already in the cache before timing (prefilled)")
    ap.add_argument("--vocab", type=int, default=8192)
    ap.add_argument("--block", type=int, default=5120)
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--heads", type=int, default=16)
    ap.add_argument("--embd", type=int, default=1536)
    ap.add_argument("--option", action="append", default=[], metavar="NAME=INT", help="omnitok_set_option switch (A/B)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the 8-stream line (B = 1 runs only)")
    a = ap.parse_args()
    V, BS, L, H, C = a.vocab, a.block, a.layers, a.heads, a.embd
    for kv in a.option:
        from omnitokenizer_amd import _lib
        _lib.set_option(kv.partition("=")[0], int(kv.partition("=")[2]))
    sd = synth_gpt_state(V, BS, L, H, C, seed=0)
    m = og.GPT(argparse.Namespace(), V, BS, n_layer=L, n_head=H, n_embd=C)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    B = a.batch
    g = torch.Generator().manual_seed(1)
    cond = torch.randint(0, V, (B, 1 + a.ctx), generator=g).cuda()
    # prefill a.ctx tokens (not timed), then time `steps` sampled tokens
    torch.manual_seed(0)
    og.sample_with_past(cond, m, 8, top_k=2048, top_p=0.9, use_graph=not a.no_graph)  # warm-up + graph capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = og.sample_with_past(cond, m, a.steps, top_k=2048, top_p=0.9, use_graph=not a.no_graph)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the prefill of ctx tokens is inside dt: measure it alone and subtract
    t1 = time.perf_counter()
    og.sample_with_past(cond, m, 1, top_k=2048, top_p=0.9, use_graph=not a.no_graph)
    torch.cuda.synchronize()
    t_prefill = time.perf_counter() - t1
    dt_sample = max(dt - t_prefill, 1e-9)
    # the bare decode step (graph replay only, no token selection) at the final context length
    idx_buf, logits_buf, replay = m.graph_step(B)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    m._pos[:B] = a.ctx + a.steps
    m._len[:B] = a.ctx + a.steps
    e0.record()
    nrep = 50
    for _ in range(nrep):
        m._pos[:B] = a.ctx + a.steps
        m._len[:B] = a.ctx + a.steps
        replay()
    e1.record()
    torch.cuda.synchronize()
    step_ms = e0.elapsed_time(e1) / nrep
    hd = C // H
    weight_bytes = (L * 12 * C * C + V * C) * 4.0
    kv_bytes = 2.0 * L * B * H * (a.ctx + a.steps) * hd * 4.0
    out_json = {
        "metric": "LM sampled tokens/sec (sample_with_past, top-k 2048 / top-p 0.9)",
        "value": round(B * a.steps / dt_sample, 1), "unit": "tokens/s", "n_gpus": 1, "batch_streams": B,
        "steps": a.steps, "ctx": a.ctx, "ms_per_token_step": round(dt_sample / a.steps * 1e3, 4),
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"GPT {L}x{C} ({H} heads, head_dim {hd}), vocab {V}, block {BS}; B={B} streams, "
                               f"{a.ctx} cached tokens + {a.steps} sampled"},
        "roofline": {"kernel": "decode step (graph replay) at the final context", "bound": "hbm",
                     "achieved": round((weight_bytes + kv_bytes) / (step_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS,
                     "unit": "GB/s", "step_ms": round(step_ms, 4), "weight_bytes": weight_bytes, "kv_bytes": kv_bytes,
                     "traffic": None},
        "kv_cache_gb": round(m.cache_bytes() / 2**30, 2),
        "prefill_ms": round(t_prefill * 1e3, 2),  # a.ctx prefix tokens (batched prefill) + 1 sampled token
    }
    out_json["roofline"]["frac"] = round(out_json["roofline"]["achieved"] / PEAK_HBM_GBS, 4)
    try:  # memory-side traffic of one step from the committed PMC pass (default model shape, B = 1 only)
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["lm_step_b1"]
        if (V, BS, L, H, C, B) == (8192, 5120, 24, 16, 1536, 1):
            out_json["roofline"]["traffic"] = pmc["read_bytes"] + pmc["write_bytes"]
            out_json["roofline"]["traffic_source"] = pmc["source"]
    except Exception:
        pass
    if B == 1 and not a.no_also:
        # the same model with 8 streams sampled together (one pass over the weights per step serves all of them)
        B8, n8 = 8, min(a.steps, 128)
        cond8 = torch.randint(0, V, (B8, 1 + a.ctx), generator=g).cuda()
        og.sample_with_past(cond8, m, 8, top_k=2048, top_p=0.9, use_graph=not a.no_graph)
        torch.cuda.synchronize()
        t8 = time.perf_counter()
        og.sample_with_past(cond8, m, n8, top_k=2048, top_p=0.9, use_graph=not a.no_graph)
        torch.cuda.synchronize()
        dt8 = time.perf_counter() - t8
        t9 = time.perf_counter()   # the prefill of the ctx prefix is inside dt8: measure it alone (+ 1 token) and subtract
        og.sample_with_past(cond8, m, 1, top_k=2048, top_p=0.9, use_graph=not a.no_graph)
        torch.cuda.synchronize()
        dt8 = max(dt8 - (time.perf_counter() - t9), 1e-9)
        # the bare decode step of the 8 streams at the final context
        _, _, replay8 = m.graph_step(B8)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(nrep):
            m._pos[:B8] = a.ctx + n8
            m._len[:B8] = a.ctx + n8
            replay8()
        e1.record()
        torch.cuda.synchronize()
        step8 = e0.elapsed_time(e1) / nrep
        kv8 = 2.0 * L * B8 * H * (a.ctx + n8) * hd * 4.0
        out_json["also"] = {"b8": {"batch_streams": B8, "steps": n8, "tokens_s": round(B8 * n8 / dt8, 1),
                                   "ms_per_token_step": round(dt8 / n8 * 1e3, 4), "step_ms": round(step8, 4),
                                   "kv_bytes": kv8, "frac_hbm": round((weight_bytes + kv8) / (step8 * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}}
    if not a.no_cpu_baseline:
        from oracle import gpt_oracle as go  # the CPU oracle is only the baseline / checker here
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        x = cond.cpu()[:1, -1:]
        with torch.no_grad():
            _, cache = go.forward_with_past(sd, cond.cpu()[:1, :min(1 + a.ctx, 64)], H, None)
            n, spent = 0, 0.0
            while n < 3 or (spent < 15.0 and n < 200):
                t = time.perf_counter()
                lg, cache = go.forward_with_past(sd, x, H, cache, position=cache[0][0].shape[2])
                spent += time.perf_counter() - t
                n += 1
        out_json["cpu_baseline"] = {"value": round(n / spent, 2), "unit": "tokens/s", "cores": torch.get_num_threads(),
                                    "kind": "port", "sample": f"{n} KV-cached oracle steps (1 stream, context "
                                                              f"{cache[0][0].shape[2]}), torch CPU fp32"}
    print(json.dumps(out_json), flush=True)


if __name__ == "__main__":
    main()
