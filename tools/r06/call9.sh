#!/bin/bash
# r06 call 9: legacy relative-position bias on the 64-query attention kernels (table in LDS) -- tests, and a stage-1 bench record A/B
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_attn_h2.py -x -q 2>&1 | tail -6 | tee $OUT/r06_attn_bias_tests.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -k "legacy or s1_" 2>&1 | tail -4 | tee -a $OUT/r06_attn_bias_tests.txt
B="--steps 10 --warmup 3 --no-clock-probe --no-cpu-baseline --no-also --stage 1 --frames 1 --batch 64"
summ() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d["kernels"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], round(d["value"]/1e6,3), "M patches/s |", " ".join(f"{n}={k[n]['ms_per_step']:.2f}({k[n].get('frac_of_mode_roof', k[n].get('frac_hbm_peak'))})" for n in sorted(k, key=lambda n:-k[n]['ms_per_step'])[:8]))
PY
}
for v in 6 3 6 3; do python bench.py $B --option attn_h2_variant=$v 2>/dev/null > $OUT/r06_stage1_v$v.json; summ $OUT/r06_stage1_v$v.json; done 2>&1 | tee $OUT/r06_stage1_ab.txt
python bench.py --steps 10 --warmup 3 --no-clock-probe --no-also --stage 1 --frames 1 --batch 64 2>/dev/null > $OUT/r06_bench_stage1.json; summ $OUT/r06_bench_stage1.json
python bench.py --steps 10 --warmup 3 --no-clock-probe --no-cpu-baseline --no-also --stage 1 --frames 17 --batch 8 2>/dev/null > $OUT/r06_stage1_vid.json; summ $OUT/r06_stage1_vid.json
