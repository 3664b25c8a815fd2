#!/bin/bash
# r06 call 18: the refined pipelined loop (6 / 5 / 10 / 11 / ROWLN 6, 7) against the plain loop (36 / 35 / 16 / 17), then the product build's tests
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
L=omnitokenizer_amd/lib/libomnitok.so
cp tools/_bin/libomnitok_meas.so $L
for rows in 1024 5120 20480; do
  echo "#### rows $rows   (6 / 5 = 128x64 / 128x128 pipelined; 36 / 35 = the same tiles, plain loop; 10, 11 tiny; out_res_ln: 6, 7 pipelined, 16, 17 plain)"
  timeout 300 tools/_bin/pl_bench --rows $rows --cfg 6,36,5,35,10,11,7,17,16,6 --iters 20 --check 2>&1 | grep -v "^   h2 ablation"
done 2>&1 | tee $OUT/r06_pl_loop1b.txt | grep -v "check: max |pl - h2\|omnitok 0.1" | sed 's/   (wg0 span.*//'
cp tools/_bin/libomnitok_prod.so $L
timeout 900 python -m pytest tests/test_gpu_gemm_pl.py -x -q 2>&1 | tail -4
python tools/latency.py --frames 1 2>&1 | grep -v amdgpu.ids
python tools/latency.py --frames 17 2>&1 | grep -v amdgpu.ids
