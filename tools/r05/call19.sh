#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05c19; mkdir -p $O
cp omnitokenizer_amd/lib/libomnitok.so /tmp/prod.so
for v in C D E; do
  cp omnitokenizer_amd/lib/variants/$v.so omnitokenizer_amd/lib/libomnitok.so
  timeout 600 python tools/r05/plt_stress.py > $O/stress_$v.txt 2>&1
  echo "== variant $v"; grep "repetitions" $O/stress_$v.txt
done
cp /tmp/prod.so omnitokenizer_amd/lib/libomnitok.so
