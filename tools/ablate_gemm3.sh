#!/bin/bash
# compile-time ablations of gemm3.hip: builds naturalspeech2_pytorch_amd/libns2hip_g3_<bits>.so for each bit set given
# (1 no in-loop DMA, 2 no MFMA, 8 no in-loop fragment reads, 64 no barrier).  usage: tools/ablate_gemm3.sh 0 1 8 9 73 2
set -e
cd "$(dirname "$0")/../naturalspeech2_pytorch_amd/csrc"
FLAGS="$G3_EXTRA --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -I../../include -mllvm -pragma-unroll-threshold=200000"
mkdir -p obj_g3
for b in "$@"; do
  ( hipcc $FLAGS -DG3_ABL=$b -c gemm3.hip -o obj_g3/gemm3_$b$G3_TAG.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../libns2hip_g3_$b$G3_TAG.so obj/gemm.o obj/gemm2.o obj_g3/gemm3_$b$G3_TAG.o obj/attention.o obj/elementwise.o obj/rvq.o obj/model_exec.o obj/capi.o ) &
done
wait
echo built "$@"
