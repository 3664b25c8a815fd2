#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; tools/_bin/mfma_chain 2>&1 | tee gpurun_out/r06_mfma_chain.txt
