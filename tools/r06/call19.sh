#!/bin/bash
# r06 call 19: thin-tile K loops against the shipped tiles (6 = 128 x 64, 5 = 128 x 128): 39 / 40 = 128 x 64 with the operands through registers (4 / 6 steps in flight), 41 = 128 x 128 likewise
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
L=omnitokenizer_amd/lib/libomnitok.so
cp tools/_bin/libomnitok_meas.so $L
for rows in 1024 2048 5120; do
  echo "#### rows $rows"
  timeout 300 tools/_bin/pl_bench --rows $rows --cfg 6,39,40,5,41,6 --iters 20 --check 2>&1 | grep -v "^   h2 ablation"
done 2>&1 | tee $OUT/r06_pl_loop3.txt | grep -v "omnitok 0.1"
cp tools/_bin/libomnitok_prod.so $L
