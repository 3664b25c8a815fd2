#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05c22; mkdir -p $O; rm -f $O/lm_*.json
for opt in $LM_ARMS; do
  timeout 120 python tools/lm_bench.py --no-cpu-baseline $(echo $opt | tr ',' '\n' | sed 's/^/--option /' | tr '\n' ' ') > $O/lm_${opt}_$RANDOM.json 2>>$O/err.txt
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05c22/lm_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["ms_per_token_step"], d["roofline"]["frac"])
    except Exception as e: print(f, "failed", e)
PY
tail -3 $O/err.txt | grep -v amdgpu
