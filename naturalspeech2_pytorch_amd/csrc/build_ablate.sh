#!/bin/bash
# experiment build with ablation switches compiled in (NS2_DBG=1|2|3 at run time): ../libns2hip_ablate.so
set -e
cd "$(dirname "$0")"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -I../../include -mllvm -pragma-unroll-threshold=200000 -DNS2_ABLATE"
mkdir -p obj_ab
for f in gemm.hip gemm2.hip attention.hip elementwise.hip rvq.hip; do hipcc $FLAGS -c $f -o obj_ab/${f%.hip}.o & done
for f in model_exec.cpp capi.cpp; do hipcc $FLAGS -x hip -c $f -o obj_ab/${f%.cpp}.o 2>/dev/null & done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libns2hip_ablate.so obj_ab/*.o
echo built ablate
