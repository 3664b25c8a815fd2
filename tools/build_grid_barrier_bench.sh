#!/bin/bash
# builds tools/_bin/grid_barrier_bench (stand-alone HIP program, no library)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/grid_barrier_bench.cpp -o tools/_bin/grid_barrier_bench
echo built tools/_bin/grid_barrier_bench
