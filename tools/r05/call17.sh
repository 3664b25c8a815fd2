#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05c17; mkdir -p $O
timeout 600 python tools/r05/plt_forensic.py > $O/forensic_heavy.txt 2>&1

tail -16 $O/forensic_heavy.txt | cut -c1-700
