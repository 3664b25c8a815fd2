#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05c20; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_lm.py -q 2>&1 | grep -v amdgpu.ids | tail -4 > $O/tests_lm.txt
for opt in "lm_ksliced=1" "lm_ksliced=0" "lm_ksliced=1" "lm_ksliced=0"; do
  timeout 120 python tools/lm_bench.py --no-cpu-baseline --option $opt > $O/lm_${opt}_$RANDOM.json 2>>$O/err.txt
done
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lmstats -- python tools/lm_bench.py --no-cpu-baseline --steps 128 > $O/lmstats.log 2>&1
cp $(ls $O/lmstats/*/*kernel_stats.csv | head -1) $O/lm_kernel_stats.csv 2>/dev/null; rm -rf $O/lmstats
tail -2 $O/tests_lm.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05c20/lm_lm_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["ms_per_token_step"], d["roofline"]["frac"])
PY
head -8 $O/lm_kernel_stats.csv | cut -c1-150
