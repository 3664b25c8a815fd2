#!/bin/bash
# r06 call 15: the pipelined K loop of the thin tiles (LOOP 1) -- bit equality, time by call size against the first form (15 / 16 / 17)
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gemm_pl.py -x -q 2>&1 | tail -5 | tee $OUT/r06_pl_tests2.txt
for rows in 1024 5120 10240 40960; do
  echo "#### rows $rows"
  timeout 300 tools/_bin/pl_bench --rows $rows --cfg 1,5,15,9,6,16,8,7,17 --iters 20 --check 2>&1 | grep -v "^   h2 ablation"
done 2>&1 | tee $OUT/r06_pl_loop1.txt | grep -v "check: max\|omnitok 0.1" | sed 's/   (wg0 span.*//'
