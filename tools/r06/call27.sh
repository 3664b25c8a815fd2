#!/bin/bash
# PL_ROWLN, two full-row workgroups per CU on the 2-stage loop of the gemm_plt kind (PlCfg LOOP_ 4): bits, launch time, C3 A/B
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_gemm_pl.py -q -k "layernorm_epilogue" 2>&1 | tail -3
python tools/r06/rowln_probe.py 1 8 6 8 1 2>&1 | grep -v amdgpu.ids | tee $OUT/r06_rowln_probe2.txt
B="--steps 6 --warmup 2 --no-clock-probe --no-cpu-baseline --no-also"
for c in 1 8 1 8; do python bench.py $B --option pl_cfg=$c 2>/dev/null > $OUT/r06_c3_rowln$c.json; python - $OUT/r06_c3_rowln$c.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d["kernels"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], " ".join(f"{n}={k[n]['ms_per_step']:.2f}" for n in ("gemm_ff_in","gemm_qkv","gemm_ff_out","gemm_out","gemm_pixels","attn_spatial")))
PY
done 2>&1 | tee $OUT/r06_c3_rowln2.txt
