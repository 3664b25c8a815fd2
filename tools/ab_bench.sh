# A/B of two builds inside one box: tools/_bin/libomnitok_base.so vs tools/_bin/libomnitok_new.so (box-to-box variation is ~2 %, within one box ~0.1 %)
L=omnitokenizer_amd/lib/libomnitok.so
for r in 1 2; do for v in base new; do cp tools/_bin/libomnitok_$v.so $L; python bench.py --steps 10 --warmup 3 --no-clock-probe 2>/dev/null > gpurun_out/ab_${v}_$r.json; python - <<PY
import json
d=json.loads(open("gpurun_out/ab_${v}_$r.json").read().strip().splitlines()[-1])
k=d["roofline"].get("families") or d.get("kernels")
print("$v $r", d["ms_per_step"], " ".join(f"{n}={k[n]['ms_per_step']:.2f}" for n in ("gemm_ff_in","gemm_qkv","gemm_ff_out","gemm_out","attn_spatial","peg3d")))
PY
done; done
cp tools/_bin/libomnitok_new.so $L
