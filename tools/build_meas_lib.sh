#!/bin/bash
# Measurement build of libomnitok.so with extra -D flags (wrong-result ablation arms) into tools/_bin/libomnitok_meas.so; the
# product library under omnitokenizer_amd/lib/ is left as it is.  bash tools/build_meas_lib.sh -DOMNITOK_ATTN_MEASUREMENT_BUILDS ...
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_bin/meas_obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -Wno-unused-result $*"
SRCS=$(python -c "from omnitokenizer_amd import build; print(' '.join(build.SOURCES))")
pids=()
for s in $SRCS; do
  o=tools/_bin/meas_obj/${s%.*}.o
  /opt/rocm/bin/hipcc $FLAGS -x hip -c omnitokenizer_amd/csrc/$s -o $o &
  pids+=($!)
  if [ ${#pids[@]} -ge 8 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libomnitok_meas.so tools/_bin/meas_obj/*.o
echo built tools/_bin/libomnitok_meas.so
