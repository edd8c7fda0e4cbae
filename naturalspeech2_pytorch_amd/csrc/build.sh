#!/bin/bash
# Build libns2hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=../libns2hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -I../../include -mllvm -pragma-unroll-threshold=200000"
mkdir -p obj
pids=()
for f in gemm.hip gemm2.hip attention.hip elementwise.hip rvq.hip backward.hip; do
  ( hipcc $FLAGS -c $f -o obj/${f%.hip}.o ) & pids+=($!)
done
for f in model_exec.cpp capi.cpp capi_train.cpp; do
  ( hipcc $FLAGS -x hip -c $f -o obj/${f%.cpp}.o ) & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT obj/gemm.o obj/gemm2.o obj/attention.o obj/elementwise.o obj/rvq.o obj/backward.o obj/model_exec.o obj/capi.o obj/capi_train.o
echo "built $(readlink -f $OUT)"
