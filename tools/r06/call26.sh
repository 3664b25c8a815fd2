#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python tools/r06/rowln_probe.py 1 6 8 2>&1 | tee gpurun_out/r06_rowln_probe.txt
