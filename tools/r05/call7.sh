#!/bin/bash
# r05 GPU call 7: LM decode with a side-stream weight prefetch into the Infinity Cache (lm_prefetch = layers ahead)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r05c7
mkdir -p $O
for pf in 0 1 2 0 1 2; do
  python tools/lm_bench.py --batch 1 --steps 256 --no-cpu-baseline --option lm_prefetch=$pf > $O/lm_b1_pf${pf}_$RANDOM.json 2>>$O/err.txt
done
python tools/lm_bench.py --batch 8 --steps 128 --no-cpu-baseline --option lm_prefetch=1 > $O/lm_b8_pf1.json 2>>$O/err.txt
python tools/lm_bench.py --batch 1 --steps 256 --no-cpu-baseline --no-graph --option lm_prefetch=1 > $O/lm_b1_pf1_nograph.json 2>>$O/err.txt
python - > $O/lm_summary.txt <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05c7/lm_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], "tok/s", d["ms_per_token_step"], "ms/token", "frac", d["roofline"]["frac"], "step_ms", d["roofline"]["step_ms"])
    except Exception as e:
        print(f, "ERR", e)
PY
python -m pytest tests/test_gpu_lm.py -q -x -k "golden or prefill or calling" 2>&1 | tail -3 > $O/tests_lm.txt
cat $O/lm_summary.txt; tail -3 $O/tests_lm.txt; tail -5 $O/err.txt
