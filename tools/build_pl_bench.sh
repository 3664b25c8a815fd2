#!/bin/bash
# builds tools/_bin/pl_bench against the in-tree libomnitok.so (rpath relative to the binary)
set -e
cd "$(dirname "$0")/.."
python omnitokenizer_amd/build.py
mkdir -p tools/_bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/pl_bench.cpp -o tools/_bin/pl_bench \
    -Lomnitokenizer_amd/lib -lomnitok -Wl,-rpath,'$ORIGIN/../../omnitokenizer_amd/lib'
echo built tools/_bin/pl_bench
