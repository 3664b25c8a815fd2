#!/bin/bash
# r06 call 5: unified data flow + tail schedule -- tests, and the sizes the verdict names (1 image, 1 clip, 8 clips, C5 B=1, C3)
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_gemm_pl.py tests/test_gpu_e2e.py tests/test_gpu_coresidency.py -x -q 2>&1 | tail -12 | tee $OUT/r06_unify_tests.txt
python tools/latency.py --frames 1 2>&1 | grep -v amdgpu.ids | tee $OUT/r06_latency.txt
python tools/latency.py --frames 17 2>&1 | grep -v amdgpu.ids | tee -a $OUT/r06_latency.txt
python tools/latency.py --frames 1 --pl-min-tokens 12288 2>&1 | grep -v amdgpu.ids | tee -a $OUT/r06_latency.txt
python tools/latency.py --frames 17 --pl-min-tokens 12288 2>&1 | grep -v amdgpu.ids | tee -a $OUT/r06_latency.txt
B="--steps 10 --warmup 3 --no-clock-probe --no-cpu-baseline --no-also"
summ() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d["kernels"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], round(d["value"]/1e6,3), "M patches/s |", " ".join(f"{n}={k[n]['ms_per_step']:.2f}({k[n].get('frac_of_mode_roof', k[n].get('frac_hbm_peak'))})" for n in sorted(k, key=lambda n:-k[n]['ms_per_step'])[:9]))
PY
}
for t in 1 0; do
  python bench.py $B --batch 8 --option pl_tail=$t 2>/dev/null > $OUT/r06_b8_tail$t.json; summ $OUT/r06_b8_tail$t.json
  python bench.py $B --batch 1 --frames 65 --resolution 512 --n-codes 16384 --option pl_tail=$t 2>/dev/null > $OUT/r06_c5_tail$t.json; summ $OUT/r06_c5_tail$t.json
done 2>&1 | tee $OUT/r06_tail_ab.txt
python bench.py $B 2>/dev/null > $OUT/r06_c3_quick.json; summ $OUT/r06_c3_quick.json | tee -a $OUT/r06_tail_ab.txt
python bench.py $B --batch 64 --frames 1 2>/dev/null > $OUT/r06_c2_quick.json; summ $OUT/r06_c2_quick.json | tee -a $OUT/r06_tail_ab.txt
