#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05c13; mkdir -p $O

timeout 300 python tools/r05/plt_check.py > $O/plt_check.txt 2>&1
tail -40 $O/plt_check.txt
