#!/bin/bash
# r06 call 10: multi-stream LM decode on the matrix cores -- tests, then lm_bench with lm_mfma 1 | 0
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_lm.py -x -q 2>&1 | tail -8 | tee $OUT/r06_lm_tests.txt
for m in 1 0 1 0; do
  timeout 200 python tools/lm_bench.py --no-cpu-baseline --option lm_mfma=$m 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lm_mfma $m', 'B=1 step_ms', d['roofline']['step_ms'], 'frac', d['roofline']['frac'], '| B=8', d['also']['b8'])"
done 2>&1 | tee $OUT/r06_lm_mfma_ab.txt
for b in 4 16; do for m in 1 0; do
  timeout 200 python tools/lm_bench.py --no-cpu-baseline --no-also --batch $b --steps 128 --option lm_mfma=$m 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lm_mfma $m B=$b', 'step_ms', d['roofline']['step_ms'], 'frac', d['roofline']['frac'], 'tokens/s', d['value'])"
done; done 2>&1 | tee -a $OUT/r06_lm_mfma_ab.txt
