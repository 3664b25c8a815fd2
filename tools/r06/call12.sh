#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gemm_pl.py tests/test_gpu_ops.py tests/test_gpu_temporal_fused.py -x -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q 2>&1 | tail -3
python tools/latency.py --frames 1 2>&1 | grep -v amdgpu.ids | tee $OUT/r06_latency2.txt
python tools/latency.py --frames 17 2>&1 | grep -v amdgpu.ids | tee -a $OUT/r06_latency2.txt
python tools/breakdown.py --frames 1 2>&1 | grep "stats_pack\|^sum" | tee -a $OUT/r06_latency2.txt
python tools/breakdown.py --frames 17 2>&1 | grep "stats_pack\|^sum" | tee -a $OUT/r06_latency2.txt
python tools/breakdown.py --frames 17 --batch 4 2>&1 | grep "stats_pack\|^sum" | tee -a $OUT/r06_latency2.txt
