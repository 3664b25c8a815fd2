#!/bin/bash
# r06 call 4: one image / one clip / 8 clips on the plane flow with the size rule of the tile family, against the fp32-activation flow
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_gemm_pl.py -x -q 2>&1 | tail -3
for spec in "1 1" "17 1" "17 2" "17 8" "1 16"; do set -- $spec; for mt in 0 12288; do echo "== frames $1 batch $2 pl_min_tokens $mt"; python - <<PY
import sys; sys.argv=["breakdown","--frames","$1","--batch","$2"]
from omnitokenizer_amd import _lib
_lib.set_option("pl_min_tokens", $mt)
sys.path.insert(0,"tools"); import breakdown; breakdown.main()
PY
done; done 2>&1 | grep -v amdgpu.ids | tee $OUT/r06_small_breakdown2.txt
