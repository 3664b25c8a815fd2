"""GPU box: timeline of ONE decode step (graph replay) from s_memtime stamps inside the K-sliced GEMV launches
(omnitok_debug_set_gemm_trace; 100 MHz counter = 10 ns): per launch the span from the first workgroup's entry to the last one's
end, the median workgroup's phases, and the gap to the next GEMV (which holds the attention launch after each qkv GEMV)."""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from omnitokenizer_amd import _lib  # noqa: E402
from omnitokenizer_amd import gpt as og  # noqa: E402
from omnitokenizer_amd.synth import synth_gpt_state  # noqa: E402

V, BS, L, H, C = 8192, 5120, 24, 16, 1536
for kv in sys.argv[1:]:
    _lib.set_option(kv.partition("=")[0], int(kv.partition("=")[2]))
sd = synth_gpt_state(V, BS, L, H, C, seed=0)
m = og.GPT(argparse.Namespace(), V, BS, n_layer=L, n_head=H, n_embd=C)
m.load_state_dict(sd, strict=True)
m = m.cuda().eval()
cond = torch.randint(0, V, (1, 1), generator=torch.Generator().manual_seed(1)).cuda()
og.sample_with_past(cond, m, 4, top_k=2048, top_p=0.9, use_graph=False)   # allocates the cache
trace = torch.zeros(256 * 1024 * 8, dtype=torch.int64, device="cuda")
_lib.load().omnitok_debug_set_gemm_trace(ctypes.c_void_p(trace.data_ptr()))
idx_buf, logits_buf, replay = m.graph_step(1)     # captured with the trace pointers baked in
_lib.load().omnitok_debug_set_gemm_trace(None)
for ctx in (256, 512):
    m._pos[:1] = ctx
    m._len[:1] = ctx
    replay()
    replay()
    torch.cuda.synchronize()
    trace.zero_()
    m._pos[:1] = ctx
    m._len[:1] = ctx
    torch.cuda.synchronize()
    replay()
    torch.cuda.synchronize()
    t = trace.cpu().view(256, 1024, 8)
    names = ["qkv", "proj", "fc1", "fc2"]
    rows = []
    for slot in range(256):
        st = t[slot, :, 0]
        live = st > 0
        if not live.any():
            continue
        a = t[slot][live]
        rows.append((len(rows), int(live.sum()), a))
    t_first = min(int(r[2][:, 0].min()) for r in rows)
    print(f"ctx {ctx}: {len(rows)} traced launches; 10 ns ticks; times in us")
    prev_end = None
    tot = {}
    for slot, nwg, a in rows:
        s0 = int(a[:, 0].min())
        e1 = int(a[:, 4].max())
        med = lambda k: float((a[:, k] - a[:, 0]).double().median()) / 100.0   # noqa: E731
        name = names[slot % 4] if slot < 96 else "head"
        gap = (s0 - prev_end) / 100.0 if prev_end is not None else 0.0
        span = (e1 - s0) / 100.0
        spread = (int(a[:, 0].max()) - s0) / 100.0
        d = tot.setdefault(name, [0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
        d[0] += 1
        for i, v in enumerate((gap, span, spread, med(1), med(2), med(3), med(4))):
            d[i + 1] += v
        if 4 <= slot < 8 or slot >= 96:
            print(f"  launch {slot:3d} {name:4s} {nwg:4d} wgs: gap before {gap:6.2f}  span {span:6.2f}  entry spread {spread:5.2f} | median wg: "
                  f"prologue done {med(1):5.2f}  first group {med(2):5.2f}  last group {med(3):5.2f}  stored {med(4):5.2f}")
        prev_end = e1
    for name, d in tot.items():
        n = d[0]
        print(f"  mean over {n:2d} {name:4s}: gap before {d[1] / n:6.2f}  span {d[2] / n:6.2f}  entry spread {d[3] / n:5.2f} | prologue {d[4] / n:5.2f} "
              f"first group {d[5] / n:5.2f} last group {d[6] / n:5.2f} stored {d[7] / n:5.2f}")
    print(f"  first entry -> last end: {(max(int(r[2][:, 4].max()) for r in rows) - t_first) / 100.0:.1f} us")
