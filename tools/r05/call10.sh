#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05c10; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --no-cpu-baseline --no-also --no-clock-probe --steps 3 --warmup 1 --option temporal_fused=1 > $O/stats.log 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats_fused.csv; rm -rf $O/stats
head -16 $O/kernel_stats_fused.csv | cut -c1-170
