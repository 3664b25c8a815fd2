#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05c15; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_temporal_fused.py -q -k "engine" 2>&1 | grep -v amdgpu.ids | tail -40 > $O/tests_unit.txt
python - > $O/engine_ab.txt 2>&1 <<'PY'
import torch
from omnitokenizer_amd import OmniTokenizer_VQGAN, _lib
from tests.helpers import GoldenCase
for name in ("heavy_s2_sdpa_r256_vid17_b8", "heavy_s2_sdpa_r256_vid17"):
    c = GoldenCase(name)
    m = OmniTokenizer_VQGAN(c.args, attention_mode=c.mode); m.load_state_dict(c.sd, strict=True); m = m.cuda().eval()
    out = {}
    for tag, tf, tk in (("unfused", 0, 1), ("gemm_pl", 1, 0), ("gemm_plt", 1, 1), ("gemm_plt 1/CU", 1, 2), ("gemm_plt again", 1, 1)):
        _lib.set_option("temporal_fused", tf); _lib.set_option("temporal_kernel", tk)
        ids, z = m.encode(c.x.cuda(), False, return_latents=True)
        out[tag] = z.clone()
        print(name, tag, "|z - golden|", float((z.cpu() - c.z).abs().max()), "|z - unfused|", float((z - out["unfused"]).abs().max()),
              "ids != golden", int((ids.cpu() != c.ids).sum()), "noise", c.fp32_noise_z, flush=True)
    _lib.set_option("temporal_fused", 1); _lib.set_option("temporal_kernel", 1)
PY
cat $O/tests_unit.txt | tail -15; cat $O/engine_ab.txt
