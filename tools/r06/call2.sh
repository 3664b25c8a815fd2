#!/bin/bash
# r06 call 2: the small-tile family of the plane GEMM -- correctness across configurations and time by call size
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gemm_pl.py tests/test_gpu_lm.py -x -q 2>&1 | tail -15 | tee $OUT/r06_pl_tests.txt
for rows in 1024 4096 5120 10240 20480 40960; do
  echo "#### rows $rows"
  timeout 300 tools/_bin/pl_bench --rows $rows --cfg 1,5,6,7 --iters 20 --check 2>&1 | grep -v "^   h2 ablation"
done 2>&1 | tee $OUT/r06_pl_small_tiles.txt | grep -v "check: max |pl - h2" | cut -c1-110
