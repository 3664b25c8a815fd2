#!/bin/bash
# GPU box: the PMC passes behind bench.py's static roofline fields (rocprofv3 --pmc, one counter group per run, with
# --kernel-trace only -- never combined with sys / hip / memory-copy traces), of the SAME command the bench line comes
# from, then tools/pmc_to_json.py turns the per-kernel means into profiles/pmc_traffic.json stamped with the sha256 of
# the csrc/ tree the library was built from.  bash tools/pmc_collect.sh <tag> [bench args...]
set -u
TAG=${1:-r06}; shift || true
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
BENCH="python bench.py --no-cpu-baseline --no-clock-probe --no-also --steps 1 --warmup 1 $*"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA SQ_INSTS_LDS"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/${TAG}_pmcraw_$name -- $BENCH > $OUT/${TAG}_pmc_$name.log 2>&1
  python tools/pmc_summary.py $OUT/${TAG}_pmcraw_$name --filter omnitok > $OUT/${TAG}_pmc_$name.csv
  rm -rf $OUT/${TAG}_pmcraw_$name
done
python tools/pmc_to_json.py $OUT/${TAG}_pmc_FETCH_SIZE.csv $OUT/${TAG}_pmc_WRITE_SIZE.csv $OUT/${TAG}_pmc_SQ_VALU_MFMA_BUSY_CYCLES.csv ${TAG} > $OUT/${TAG}_pmc_traffic.json
head -c 600 $OUT/${TAG}_pmc_traffic.json
