#!/bin/bash
# Round-end evidence in one GPU call: full GPU suite, the default bench line, the LM bench line, rocprofv3 kernel stats
# of the bench command and of the LM decode loop.  bash tools/final_round.sh <tag>   (outputs under gpurun_out/<tag>_*)
set -u
TAG=${1:-r06}
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp; export TMPDIR=/tmp; cd "$ROOT"
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -s 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_gpu_tests.txt; tail -1 $OUT/${TAG}_gpu_tests.txt
# the parity result lines the tests print (heavy-statistics fixtures per arithmetic mode, all 32 clips of the bench batch, C5 stress)
grep -E "id flips|vs the oracle|stress" $OUT/${TAG}_gpu_tests.txt > $OUT/${TAG}_gpu_parity_prints.txt
timeout 200 python bench.py > $OUT/${TAG}_bench_c3.json 2> $OUT/${TAG}_bench_c3.err; cut -c1-300 $OUT/${TAG}_bench_c3.json
timeout 200 python bench.py --frames 1 --batch 64 --no-cpu-baseline > $OUT/${TAG}_bench_c2.json 2> $OUT/${TAG}_bench_c2.err; cut -c1-200 $OUT/${TAG}_bench_c2.json
timeout 200 python bench.py --frames 65 --resolution 512 --n-codes 16384 --batch 1 --no-cpu-baseline > $OUT/${TAG}_bench_c5.json 2> $OUT/${TAG}_bench_c5.err; cut -c1-200 $OUT/${TAG}_bench_c5.json
timeout 200 python bench.py --stage 1 --frames 1 --batch 64 --no-cpu-baseline --no-also > $OUT/${TAG}_bench_stage1.json 2> $OUT/${TAG}_bench_stage1.err; cut -c1-200 $OUT/${TAG}_bench_stage1.json
timeout 200 python bench.py --batch 8 --no-cpu-baseline --no-also > $OUT/${TAG}_bench_8clips.json 2> $OUT/${TAG}_bench_8clips.err; cut -c1-200 $OUT/${TAG}_bench_8clips.json
{ python tools/latency.py --frames 1; python tools/latency.py --frames 17; } 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_latency.txt; cat $OUT/${TAG}_latency.txt
timeout 120 python tools/lm_bench.py > $OUT/${TAG}_lm_bench.json 2> $OUT/${TAG}_lm_bench.err; cut -c1-300 $OUT/${TAG}_lm_bench.json
timeout 300 python tools/lm_bench.py --ctx 4608 --steps 512 --no-cpu-baseline > $OUT/${TAG}_lm_bench_ctx4608.json 2> $OUT/${TAG}_lm_bench_ctx4608.err; cut -c1-300 $OUT/${TAG}_lm_bench_ctx4608.json
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python bench.py --no-cpu-baseline --no-also --steps 3 --warmup 1 > $OUT/${TAG}_stats_bench.log 2>&1
cp $(ls $OUT/${TAG}_stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats.csv 2>/dev/null; rm -rf $OUT/${TAG}_stats
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_lmstats -- python tools/lm_bench.py --no-cpu-baseline --steps 128 > $OUT/${TAG}_lmstats.log 2>&1
cp $(ls $OUT/${TAG}_lmstats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_lm_kernel_stats.csv 2>/dev/null; rm -rf $OUT/${TAG}_lmstats
head -8 $OUT/${TAG}_kernel_stats.csv | cut -c1-160
