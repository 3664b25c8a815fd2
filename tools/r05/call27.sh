#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 150 python tools/lm_bench.py > $OUT/r05_lm_bench.json 2> $OUT/r05_lm_bench.err; cut -c1-400 $OUT/r05_lm_bench.json
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r05_lmstats -- python tools/lm_bench.py --no-cpu-baseline --no-also --steps 128 > $OUT/r05_lmstats.log 2>&1
cp $(ls $OUT/r05_lmstats/*/*kernel_stats.csv | head -1) $OUT/r05_lm_kernel_stats.csv 2>/dev/null; rm -rf $OUT/r05_lmstats
bash tools/pmc_collect.sh r05 > $OUT/r05_pmc_collect.log 2>&1; tail -3 $OUT/r05_pmc_collect.log | cut -c1-200
