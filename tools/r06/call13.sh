#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
B="--steps 10 --warmup 3 --no-clock-probe --no-cpu-baseline --no-also"
summ() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d["kernels"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], round(d["value"]/1e6,3), "M patches/s |", " ".join(f"{n}={k[n]['ms_per_step']:.2f}" for n in sorted(k, key=lambda n:-k[n]['ms_per_step'])[:9]))
PY
}
for t in 1 100000 1 100000; do
  python bench.py $B --batch 8 --option sp_small_blocks=$t 2>/dev/null > $OUT/r06_b8_sp$t.json; summ $OUT/r06_b8_sp$t.json
done
for t in 1 100000; do python bench.py $B --option sp_small_blocks=$t 2>/dev/null > $OUT/r06_c3_sp$t.json; summ $OUT/r06_c3_sp$t.json; done
for t in 1 100000; do python bench.py $B --batch 1 --frames 65 --resolution 512 --n-codes 16384 --option sp_small_blocks=$t 2>/dev/null > $OUT/r06_c5_sp$t.json; summ $OUT/r06_c5_sp$t.json; done
