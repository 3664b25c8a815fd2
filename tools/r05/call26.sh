#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05c26; mkdir -p $O; rm -f $O/*.json
timeout 600 python -m pytest tests/test_gpu_lm.py -q -x 2>&1 | grep -v amdgpu.ids | tail -4
for opt in "lm_ksliced=2" "lm_ksliced=1"; do
  timeout 120 python tools/lm_bench.py --no-cpu-baseline --option $opt > $O/lm_${opt}.json 2>>$O/err.txt
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05c26/lm_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["ms_per_token_step"], d["roofline"]["frac"], d.get("also"))
PY
