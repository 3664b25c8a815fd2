#!/bin/bash
# Round-end evidence in one GPU call: full GPU suite, the default bench line, the LM bench line, rocprofv3 kernel stats
# of the bench command and of the LM decode loop.  bash tools/final_round.sh <tag>   (outputs under gpurun_out/<tag>_*)
set -u
TAG=${1:-r02_final}
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp; export TMPDIR=/tmp; cd "$ROOT"
OUT=gpurun_out; mkdir -p $OUT
timeout 400 python -m pytest tests -q -m gpu -s 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_gpu_tests.txt; tail -1 $OUT/${TAG}_gpu_tests.txt
timeout 200 python bench.py > $OUT/${TAG}_bench_c3.json 2> $OUT/${TAG}_bench_c3.err; cut -c1-300 $OUT/${TAG}_bench_c3.json
timeout 120 python tools/lm_bench.py > $OUT/${TAG}_lm_bench.json 2> $OUT/${TAG}_lm_bench.err; cut -c1-300 $OUT/${TAG}_lm_bench.json
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/${TAG}_stats_bench.log 2>&1
cp $(ls $OUT/${TAG}_stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats.csv 2>/dev/null; rm -rf $OUT/${TAG}_stats
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_lmstats -- python tools/lm_bench.py --no-cpu-baseline --steps 128 > $OUT/${TAG}_lmstats.log 2>&1
cp $(ls $OUT/${TAG}_lmstats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_lm_kernel_stats.csv 2>/dev/null; rm -rf $OUT/${TAG}_lmstats
head -8 $OUT/${TAG}_kernel_stats.csv | cut -c1-160
