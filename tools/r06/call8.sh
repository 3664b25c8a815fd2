#!/bin/bash
# LM decode: the round's base build (packed fp32 on) against the current build (no packed fp32, row offset in voffset)
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
for r in 1 2; do for v in base cur; do
  if [ $v = base ]; then export OMNITOK_LIB=$PWD/tools/_bin/libomnitok_base.so; else unset OMNITOK_LIB; fi
  timeout 200 python tools/lm_bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v $r', 'B=1 step_ms', d['roofline']['step_ms'], 'frac', d['roofline']['frac'], 'tokens/s', d['value'], '| B=8', d['also']['b8'])"
done; done 2>&1 | tee $OUT/r06_lm_ab.txt
unset OMNITOK_LIB
timeout 300 python tools/lm_bench.py --ctx 4608 --steps 512 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/r06_lm_ctx4608.json; cut -c1-900 $OUT/r06_lm_ctx4608.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
