#!/bin/bash
# r05 GPU call 1: new-kernel tests, temporal_chunk sweep (same box), C2 with the one-plane PEG kernel, LM baseline
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r05c1
mkdir -p $O
python -m pytest tests/test_gpu_ops.py -q -x -k "peg or prevq or layernorm or pre_vq" 2>&1 | tail -8 > $O/tests_ops.txt
python -m pytest tests/test_gpu_e2e.py -q -x -k "prevq_fusion or small_calls or (encode_decode_vs_reference_golden and r256)" 2>&1 | tail -8 > $O/tests_e2e.txt
fam() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d.get("kernels") or {}
names=("gemm_ff_in","gemm_qkv","gemm_ff_out","gemm_out","attn_spatial","attn_temporal","peg3d","stats_pack","layernorm","pre_vq","vq_argmin")
print(sys.argv[1], d["ms_per_step"], " ".join(f"{n}={k[n]['ms_per_step']:.3f}" for n in names if n in k))
PY
}
for ch in 0 4 6 8 2 0 4; do
  python bench.py --steps 10 --warmup 3 --no-clock-probe --no-also --no-cpu-baseline --option temporal_chunk=$ch > $O/c3_chunk${ch}_$RANDOM.json 2>$O/err.txt
done
for f in $O/c3_chunk*.json; do fam $f; done > $O/chunk_sweep.txt
for v in 1 2 1 2; do
  python bench.py --frames 1 --batch 64 --steps 10 --warmup 3 --no-clock-probe --no-also --no-cpu-baseline --option peg_variant=$v > $O/c2_peg${v}_$RANDOM.json 2>>$O/err.txt
done
for f in $O/c2_peg*.json; do fam $f; done > $O/c2_peg.txt
python bench.py --steps 10 --warmup 3 --no-clock-probe --no-also --no-cpu-baseline --option prevq_fuse=0 > $O/c3_prevq0.json 2>>$O/err.txt
fam $O/c3_prevq0.json >> $O/chunk_sweep.txt
python tools/lm_bench.py --batch 1 --steps 256 > $O/lm_b1.json 2>>$O/err.txt
tail -3 $O/*.txt
