#!/bin/bash
# r06 call 17: what bounds a thin tile's K step?  ablation arms of the 128 x 64 configuration (measurement build: wrong results)
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
for rows in 1024; do
  echo "#### rows $rows  (6 = shipped; 21 no vmcnt wait; 22 no barrier; 23 neither; 24 no DMA in the K loop; 27 MFMA + fragment reads only; 28 no epilogue; 36 pipelined loop; 8 deep ring)"
  timeout 300 tools/_bin/pl_bench --rows $rows --cfg 6,21,22,23,24,27,28,36,8,6 --iters 20 --shape q_or_out 2>&1 | grep -v "^   h2"
  timeout 300 tools/_bin/pl_bench --rows $rows --cfg 6,21,22,23,24,27,28,36,8,6 --iters 20 --shape ff_out 2>&1 | grep -v "^   h2"
done 2>&1 | tee $OUT/r06_pl_thin_ablation.txt
