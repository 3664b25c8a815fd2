#!/bin/bash
# r05 GPU call 5: fused LayerNorm+pre_vq after the explicit row_stats, screened VQ in the engine, LM K-split grids
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r05c5
mkdir -p $O
python tools/r05/prevq_debug.py 2>&1 | grep -v amdgpu.ids | head -12 > $O/prevq_debug.txt
python -m pytest tests/test_gpu_ops.py -q -k "prevq or pre_vq or vq_ or layernorm or stats" 2>&1 | tail -8 > $O/tests_ops.txt
python -m pytest tests/test_gpu_e2e.py -q -k "prevq_fusion or mutates or ckpt_parity or forward_codebook or (encode_decode_vs_reference_golden and (r256 or r64))" 2>&1 | tail -8 > $O/tests_e2e.txt
python -m pytest tests/test_gpu_lm.py -q -x 2>&1 | tail -8 > $O/tests_lm.txt
fam() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d.get("kernels") or {}
names=("gemm_ff_in","gemm_qkv","gemm_ff_out","gemm_out","attn_spatial","attn_temporal","peg3d","stats_pack","layernorm","pre_vq","vq_argmin")
print(sys.argv[1], d["ms_per_step"], " ".join(f"{n}={k[n]['ms_per_step']:.3f}" for n in names if n in k))
PY
}
for opt in "vq_screen=1" "vq_screen=0" "vq_screen=1"; do
  python bench.py --steps 10 --warmup 3 --no-clock-probe --no-also --no-cpu-baseline --option $opt > $O/c3_${opt}_$RANDOM.json 2>>$O/err.txt
done
for f in $O/c3_*.json; do fam $f; done > $O/c3_ab.txt
for ks in 1 0 1 0; do
  OMNITOK_LM_KSPLIT=$ks python tools/lm_bench.py --batch 1 --steps 256 --option lm_ksplit=$ks > $O/lm_b1_ks${ks}_$RANDOM.json 2>>$O/err.txt
done
python tools/lm_bench.py --batch 8 --steps 128 > $O/lm_b8.json 2>>$O/err.txt
python - > $O/lm_summary.txt <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05c5/lm_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], "tok/s", d["ms_per_token_step"], "ms/token", "frac", d["roofline"]["frac"], "step_ms", d["roofline"]["step_ms"])
    except Exception as e:
        print(f, "ERR", e)
PY
tail -n 5 $O/*.txt
