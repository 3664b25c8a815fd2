#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05c21; mkdir -p $O
timeout 120 python tools/r05/boundary_probe.py > $O/boundary.txt 2>&1
env | grep -i "HIP\|HSA\|ROC\|AMD" > $O/env.txt
cat $O/boundary.txt | grep -v amdgpu; cat $O/env.txt | head -20
