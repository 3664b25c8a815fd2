#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
( time python bench.py ) > $OUT/r06_bench_default.json 2> $OUT/r06_bench_default.err; tail -4 $OUT/r06_bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_bench_default.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"])
print("parity", json.dumps(d["parity"]))
print("cpu", json.dumps(d["cpu_baseline"])[:300])
a=d["also"]
print("also", {k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ("ms_per_step","patches_s","flips_all_clips","error")}) for k,v in a.items() if k!="parity_heavy"})
print("heavy", json.dumps(a.get("parity_heavy"))[:900])
PY
