#!/bin/bash
# experiment builds of gemm2.hip: tools/ablate_gemm2.sh <tag> <extra flags...>  -> libns2hip_g2_<tag>.so
set -e
TAG=$1; shift
cd "$(dirname "$0")/../naturalspeech2_pytorch_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -I../../include -mllvm -pragma-unroll-threshold=200000"
mkdir -p obj_g3
hipcc $FLAGS "$@" -c gemm2.hip -o obj_g3/gemm2_$TAG.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libns2hip_g2_$TAG.so obj/gemm.o obj_g3/gemm2_$TAG.o obj/attention.o obj/elementwise.o obj/rvq.o obj/model_exec.o obj/capi.o
echo built $TAG
