#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
L=omnitokenizer_amd/lib/libomnitok.so
cp tools/_bin/libomnitok_meas.so $L
for rows in 1024; do
  echo "#### rows $rows  (6 plain R4; 36 pipelined R4 (3 steps ahead); 8 pipelined R8 (7 ahead); 37 / 38 barrier-per-2 R8 / R6; 39 / 40 registers 4 / 6 ahead)"
  timeout 300 tools/_bin/pl_bench --rows $rows --cfg 6,36,8,37,38,39,40,6 --iters 20 --shape q_or_out 2>&1 | grep -v "^   h2\|omnitok 0.1"
  timeout 300 tools/_bin/pl_bench --rows $rows --cfg 6,36,8,37,38,39,40,6 --iters 20 --shape ff_out 2>&1 | grep -v "^   h2\|omnitok 0.1"
done 2>&1 | tee $OUT/r06_pl_loops_1024.txt
cp tools/_bin/libomnitok_prod.so $L
