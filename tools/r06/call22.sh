#!/bin/bash
# C3 with every plane GEMM forced to one tile configuration ("pl_cfg" 1 | 5 | 2): do the heavy-epilogue launches (q|k pack, V pack) gain from two workgroups per CU?
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
B="--steps 6 --warmup 2 --no-clock-probe --no-cpu-baseline --no-also"
for c in 1 5 2 1; do python bench.py $B --option pl_cfg=$c 2>/dev/null > $OUT/r06_c3_plcfg$c.json; python - $OUT/r06_c3_plcfg$c.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d["kernels"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], " ".join(f"{n}={k[n]['ms_per_step']:.2f}" for n in ("gemm_ff_in","gemm_qkv","gemm_ff_out","gemm_out","gemm_pixels","attn_spatial")))
PY
done 2>&1 | tee $OUT/r06_c3_plcfg.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/cfg5prof -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-clock-probe --no-cpu-baseline --no-also --option pl_cfg=5 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(ls $OUT/cfg5prof/*/*kernel_stats.csv | head -1); head -12 $f | cut -c1-120,121-175 > $OUT/r06_c3_plcfg5_kernel_stats.txt; rm -rf $OUT/cfg5prof; cat $OUT/r06_c3_plcfg5_kernel_stats.txt | cut -c1-175
