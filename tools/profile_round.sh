#!/bin/bash
# rocprofv3 evidence for the round: kernel stats of the bench command, PMC passes (FETCH_SIZE / WRITE_SIZE /
# MFMA busy in separate runs).  Run on the GPU box: bash tools/profile_round.sh <tag> [bench args...]
# Outputs under gpurun_out/<tag>_*: copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r02}; shift || true
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
BENCH="python bench.py --no-cpu-baseline $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- $BENCH --steps 3 --warmup 1 > $OUT/${TAG}_stats_bench.log 2>&1
cp $(ls $OUT/${TAG}_stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats.csv 2>/dev/null
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$name -- $BENCH --steps 1 --warmup 1 > $OUT/${TAG}_pmc_$name.log 2>&1
  python tools/pmc_summary.py $OUT/${TAG}_pmc_$name --filter omnitok > $OUT/${TAG}_pmc_$name.csv
  rm -rf $OUT/${TAG}_pmc_$name
done
rm -rf $OUT/${TAG}_stats
tail -1 $OUT/${TAG}_stats_bench.log | cut -c1-400
head -12 $OUT/${TAG}_kernel_stats.csv | cut -c1-200
