#!/bin/bash
# r06 call 1: (a) cost of the no-packed-fp32 build at C3 (A/B inside one box), (b) the new co-residency stress tests + LM tests,
# (c) per-family time of one image / one clip on both data flows (the baseline the small-tile GEMM family is measured against)
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
L=omnitokenizer_amd/lib/libomnitok.so
for r in 1 2; do for v in base new; do cp tools/_bin/libomnitok_$v.so $L
  python bench.py --steps 10 --warmup 3 --no-clock-probe --no-cpu-baseline --no-also 2>/dev/null > $OUT/ab_${v}_$r.json
  python - <<PY
import json
d=json.loads(open("$OUT/ab_${v}_$r.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("$v $r", d["ms_per_step"], " ".join(f"{n}={k[n]['ms_per_step']:.2f}" for n in sorted(k, key=lambda n:-k[n]['ms_per_step'])[:14]))
PY
done; done 2>&1 | tee $OUT/r06_ab_no_packed.txt
cp tools/_bin/libomnitok_new.so $L
timeout 900 python -m pytest tests/test_gpu_coresidency.py -x -q 2>&1 | tail -15 | tee $OUT/r06_coresidency.txt
timeout 900 python -m pytest tests/test_gpu_lm.py tests/test_gpu_temporal_fused.py -x -q 2>&1 | tail -5
for fr in 1 17; do for mt in 0 12288; do echo "== frames $fr pl_min_tokens $mt"; python - <<PY
import sys; sys.argv=["breakdown","--frames","$fr"]
from omnitokenizer_amd import _lib
_lib.set_option("pl_min_tokens", $mt)
sys.path.insert(0,"tools"); import breakdown; breakdown.main()
PY
done; done 2>&1 | tee $OUT/r06_small_breakdown.txt
