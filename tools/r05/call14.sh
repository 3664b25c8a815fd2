#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05c14; mkdir -p $O
timeout 300 python tools/r05/plt_check.py > $O/plt_check.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_temporal_fused.py -q 2>&1 | grep -v amdgpu.ids | tail -5 > $O/tests_unit.txt
for opt in "temporal_kernel=1" "temporal_kernel=0" "temporal_kernel=1" "temporal_kernel=0"; do
  timeout 200 python bench.py --steps 10 --warmup 3 --no-clock-probe --no-also --no-cpu-baseline --option $opt > $O/c3_${opt}_$RANDOM.json 2>>$O/err.txt
done
python - <<'PY' > $O/c3_ab.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/r05c14/c3_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["kernels"]
    print(f, d["ms_per_step"], " ".join(f"{n}={k[n]['ms_per_step']:.3f}" for n in ("gemm_qkv","attn_temporal","stats_pack","gemm_out") if n in k))
PY
tail -30 $O/plt_check.txt; tail -3 $O/tests_unit.txt; cat $O/c3_ab.txt
