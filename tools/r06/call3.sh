#!/bin/bash
# r06 call 3: deep-ring thin tiles -- correctness, time by call size, and the C3 step against the round's base build
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gemm_pl.py tests/test_gpu_lm.py -x -q 2>&1 | tail -5 | tee $OUT/r06_pl_tests.txt
for rows in 1024 5120 10240 20480 40960; do
  echo "#### rows $rows"
  timeout 300 tools/_bin/pl_bench --rows $rows --cfg 1,5,9,6,8,7 --iters 20 --check 2>&1 | grep -v "^   h2 ablation"
done 2>&1 | tee $OUT/r06_pl_small_tiles2.txt | grep -v "check: max |pl - h2" | cut -c1-64
L=omnitokenizer_amd/lib/libomnitok.so
cp $L tools/_bin/libomnitok_cur.so
for r in 1 2; do for v in base cur; do cp tools/_bin/libomnitok_$v.so $L
  python bench.py --steps 10 --warmup 3 --no-clock-probe --no-cpu-baseline --no-also 2>/dev/null > $OUT/ab_${v}_$r.json
  python - <<PY
import json
d=json.loads(open("$OUT/ab_${v}_$r.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("$v $r", d["ms_per_step"], " ".join(f"{n}={k[n]['ms_per_step']:.2f}" for n in sorted(k, key=lambda n:-k[n]['ms_per_step'])[:12]))
PY
done; done 2>&1 | tee $OUT/r06_ab_cur.txt
cp tools/_bin/libomnitok_cur.so $L
