#!/bin/bash
# r06 call 16: tiny tiles (128 x 32, 64 x 32) -- bit equality, time at 1024 / 2048 / 5120 rows, one-image / one-clip breakdown
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gemm_pl.py -x -q 2>&1 | tail -5 | tee $OUT/r06_pl_tests3.txt
for rows in 1024 2048 5120; do
  echo "#### rows $rows"
  timeout 300 tools/_bin/pl_bench --rows $rows --cfg 6,10,11,0,7 --iters 20 --check 2>&1 | grep -v "^   h2 ablation"
done 2>&1 | tee $OUT/r06_pl_tiny.txt | grep -v "check: max\|omnitok 0.1" | sed 's/   (wg0 span.*//'
python tools/latency.py --frames 1 2>&1 | grep -v amdgpu.ids
python tools/latency.py --frames 17 2>&1 | grep -v amdgpu.ids
python tools/breakdown.py --frames 1 2>&1 | grep -v amdgpu.ids | head -8
