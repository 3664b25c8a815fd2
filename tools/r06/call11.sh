#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_lm.py -x -q -k "gemv" 2>&1 | tail -3
for o in "lm_mfma=1" "lm_mfma=1 --option lm_mfma_mult=2" "lm_mfma=0"; do
  timeout 200 python tools/lm_bench.py --no-cpu-baseline --option $o 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$o', 'B=1 step_ms', d['roofline']['step_ms'], '| B=8', d['also']['b8']['step_ms'], d['also']['b8']['tokens_s'])"
done 2>&1 | tee $OUT/r06_lm_mfma_ab2.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/lmprof -- python $GRAFT_REPO_ROOT/tools/lm_bench.py --no-cpu-baseline --batch 8 --steps 64 --no-also > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(ls $OUT/lmprof/*/*kernel_stats.csv | head -1); head -14 $f | cut -c1-200 | tee $OUT/r06_lm_b8_kernel_stats.txt; rm -rf $OUT/lmprof
