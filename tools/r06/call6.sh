#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 300 python tools/r06/tail_debug.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r06_tail_debug.txt
for rows in 163840; do
  echo "#### rows $rows"
  timeout 300 tools/_bin/pl_bench --rows $rows --cfg 1,5,6,7,1 --iters 10 2>&1 | grep -v "^   h2 ablation"
done 2>&1 | tee $OUT/r06_pl_c3_scale.txt | cut -c1-70
