#!/bin/bash
# r05 GPU call 8: fused temporal stage (PL_TSCORE / PL_TPV): unit tests, e2e parity, C3 A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r05c8
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_temporal_fused.py -q -s 2>&1 | grep -v amdgpu.ids | tail -30 > $O/tests_unit.txt
timeout 600 python -m pytest tests/test_gpu_e2e.py -q -s -k "vid or heavy or c3_batch or full_size or prevq_fusion" 2>&1 | grep -v amdgpu.ids | tail -40 > $O/tests_e2e.txt
fam() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d.get("kernels") or {}
names=("gemm_ff_in","gemm_qkv","gemm_ff_out","gemm_out","attn_spatial","attn_temporal","peg3d","stats_pack","layernorm","pre_vq","vq_argmin")
print(sys.argv[1], d["ms_per_step"], " ".join(f"{n}={k[n]['ms_per_step']:.3f}" for n in names if n in k))
PY
}
for opt in "temporal_fused=1" "temporal_fused=0" "temporal_fused=1" "temporal_fused=0"; do
  timeout 200 python bench.py --steps 10 --warmup 3 --no-clock-probe --no-also --no-cpu-baseline --option $opt > $O/c3_${opt}_$RANDOM.json 2>>$O/err.txt
done
for f in $O/c3_*.json; do fam $f; done > $O/c3_ab.txt
tail -n 12 $O/tests_unit.txt; tail -n 6 $O/tests_e2e.txt; cat $O/c3_ab.txt
