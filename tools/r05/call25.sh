#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05c25; mkdir -p $O
timeout 600 python tools/r05/plt_stress.py > $O/stress_$RANDOM.txt 2>&1
grep "repetitions" $O/stress_*.txt | tail -3
timeout 300 python -m pytest tests/test_gpu_temporal_fused.py tests/test_gpu_lm.py -q 2>&1 | grep -v amdgpu.ids | tail -2
rocm-smi --showserial 2>/dev/null | grep -i serial | head -2
