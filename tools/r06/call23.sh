#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -k "tile_schedule or one_data_flow or full_size" 2>&1 | tail -6
