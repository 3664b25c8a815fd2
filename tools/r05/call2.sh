#!/bin/bash
# r05 GPU call 2: fused LayerNorm + pre_vq (rewritten), screened VQ search, encode() state mutation, checkpoint tool -- tests + timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r05c3
mkdir -p $O
python tools/r05/prevq_debug.py > $O/prevq_debug.txt 2>&1; python -m pytest tests/test_gpu_ops.py -q -k "prevq or pre_vq or vq_ or peg or layernorm" 2>&1 | tail -15 > $O/tests_ops.txt
python -m pytest tests/test_gpu_e2e.py -q -k "prevq_fusion or mutates or ckpt_parity or forward_codebook or forward_log or (encode_decode_vs_reference_golden and (r256 or r64))" 2>&1 | tail -15 > $O/tests_e2e.txt
python - > $O/vq_timing.txt 2>&1 <<'PY'
import torch, numpy as np, time
from omnitokenizer_amd import ops, _lib
torch.manual_seed(0)
def bench(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for n, nc in ((163840, 8192), (65536, 8192), (69632, 16384), (8 * 69632, 16384), (5120, 8192), (1024, 8192)):
    E = torch.randn(nc, 8, device="cuda")
    z = torch.nn.functional.normalize(torch.randn(n, 8, device="cuda"), dim=1)
    prep = ops.vq_prepare(E)
    scr = ops.vq_screen_prepare(E, prep[1])
    ex = bench(lambda: ops.vq_argmin(z, E, prep))
    line = f"n {n} codes {nc}: exact {ex:.1f} us"
    for sp in (0, 1, 2, 4, 8, 16):
        _lib.set_option("vq_screen_split", sp)
        t = bench(lambda: ops.vq_argmin_screened(z, E, prep, scr))
        line += f" | screened split {sp}: {t:.1f}"
    _lib.set_option("vq_screen_split", 0)
    same = torch.equal(ops.vq_argmin(z, E, prep), ops.vq_argmin_screened(z, E, prep, scr))
    print(line, "| equal", same, flush=True)
PY
fam() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d.get("kernels") or {}
names=("gemm_ff_in","gemm_qkv","gemm_ff_out","gemm_out","attn_spatial","attn_temporal","peg3d","stats_pack","layernorm","pre_vq","vq_argmin")
print(sys.argv[1], d["ms_per_step"], " ".join(f"{n}={k[n]['ms_per_step']:.3f}" for n in names if n in k))
PY
}
for opt in "prevq_fuse=1" "prevq_fuse=0" "vq_screen=0" "prevq_fuse=1"; do
  python bench.py --steps 10 --warmup 3 --no-clock-probe --no-also --no-cpu-baseline --option $opt > $O/c3_${opt}_$RANDOM.json 2>>$O/err.txt
done
for f in $O/c3_*.json; do fam $f; done > $O/c3_ab.txt
tail -n 4 $O/*.txt
