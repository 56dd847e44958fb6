#!/bin/bash
# Build the libraries of the summation-grouping experiment (NOTEBOOK 11.2): round 5's rendering of the per-edge product
# with round 6's grouping switched in for the gradient (1), the product (2), the cost (4) -> tools/exp/_bisect/lib_<m>.so
# Only the strict translation unit differs; the other objects are the product's (graphik_amd/lib/obj).
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
python -m graphik_amd.build >/dev/null
mkdir -p "$R/tools/exp/_bisect"
for m in ${MODES:-0 1 2 3 4 5 6 7}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$R/include" -I"$R/graphik_amd/csrc" -I"$R/tools/exp" \
    -DGIK_STRICT_EXP=$m '-DGIK_STRICT_HEADER="gik_wave_strict_bisect.hip.h"' \
    -c "$R/graphik_amd/csrc/gik_k_wave3_strict.hip" -o "$R/tools/exp/_bisect/strict_$m.o" &
done
wait
for m in ${MODES:-0 1 2 3 4 5 6 7}; do
  objs=$(ls "$R"/graphik_amd/lib/obj/*.o | grep -v gik_k_wave3_strict)
  hipcc --offload-arch=gfx950 -shared -fPIC $objs "$R/tools/exp/_bisect/strict_$m.o" -o "$R/tools/exp/_bisect/lib_$m.so"
  rm "$R/tools/exp/_bisect/strict_$m.o"
done
ls -la "$R/tools/exp/_bisect"
