#!/bin/bash
# GPU box: correctness + timing + PMC passes of the GEMM arms (pl cfgs, h2 and its ablation builds).
# usage: bash tools/pl_probe.sh <tag>
set -u
TAG=${1:-r03}
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
B=tools/_bin/pl_bench
timeout 300 $B --check --iters 3 > $OUT/${TAG}_pl_check.txt 2>&1; echo "check rc=$?" >> $OUT/${TAG}_pl_check.txt
timeout 300 $B --iters 20 --h2dbg 2,14 > $OUT/${TAG}_pl_bench.txt 2>&1
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f2)
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/${TAG}_plpmc_$name -- $B --iters 2 --h2dbg 2,14 --shape qkv > $OUT/${TAG}_plpmc_$name.log 2>&1
  python tools/pmc_summary.py $OUT/${TAG}_plpmc_$name --filter gemm_ > $OUT/${TAG}_plpmc_qkv_$name.csv
  rm -rf $OUT/${TAG}_plpmc_$name
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/${TAG}_plpmc_$name -- $B --iters 2 --shape ff_in > $OUT/${TAG}_plpmc_$name.log 2>&1
  python tools/pmc_summary.py $OUT/${TAG}_plpmc_$name --filter gemm_ > $OUT/${TAG}_plpmc_ffin_$name.csv
  rm -rf $OUT/${TAG}_plpmc_$name
done
cat $OUT/${TAG}_pl_check.txt | tail -30
cat $OUT/${TAG}_pl_bench.txt
