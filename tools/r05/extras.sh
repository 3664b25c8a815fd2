#!/bin/bash
# r05 evidence extras: the checkpoint-parity tool on a synthetic heavy-tailed PL checkpoint (17-frame 256^2 clips, both arithmetic
# modes, oracle as the checker on the GPU box), small-call latency table, 8 clips / 8 C5 clips throughput
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05x; mkdir -p $O
python - > $O/make_ckpt.txt 2>&1 <<'PY'
import torch
from omnitokenizer_amd import make_args, synth
from omnitokenizer_amd.config import OmniTokConfig
args = make_args(2, resolution=256)
sd = synth.synth_state_dict(OmniTokConfig.from_args(args), seed=0, profile="heavy")
sd["image_discriminator.blocks.0.weight"] = torch.zeros(4, 4)
torch.save({"state_dict": sd, "hyper_parameters": {"args": args}}, "/tmp/synth_heavy.ckpt")
print("wrote /tmp/synth_heavy.ckpt", len(sd), "tensors")
PY
timeout 600 python tools/ckpt_parity.py --ckpt /tmp/synth_heavy.ckpt --synthetic 2 --frames 17 --batch 2 2>/dev/null | grep -v amdgpu.ids > $O/ckpt_parity_demo.jsonl
timeout 300 python tools/latency.py --frames 1 --batch 1 2>/dev/null | grep -v amdgpu.ids > $O/latency_1img.txt
timeout 300 python tools/latency.py --frames 17 --batch 1 2>/dev/null | grep -v amdgpu.ids > $O/latency_1clip.txt
timeout 300 python bench.py --batch 8 --steps 10 --warmup 3 --no-also --no-cpu-baseline --no-clock-probe > $O/bench_8clips.json 2>/dev/null
timeout 300 python bench.py --frames 65 --resolution 512 --n-codes 16384 --batch 8 --steps 3 --warmup 1 --no-also --no-cpu-baseline --no-clock-probe > $O/bench_c5_b8.json 2>/dev/null
tail -3 $O/ckpt_parity_demo.jsonl | cut -c1-400; cat $O/latency_1img.txt $O/latency_1clip.txt | tail -6; cut -c1-200 $O/bench_8clips.json; cut -c1-200 $O/bench_c5_b8.json
