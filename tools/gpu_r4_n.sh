#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4n; mkdir -p $O
L=$PWD/naturalspeech2_pytorch_amd/libns2hip_noslp.so
for rep in 1 2 3; do
  NS2_LIB=$L python tools/bench_attention.py >> $O/att_noslp.txt 2>/dev/null
  python tools/bench_attention.py >> $O/att_slp.txt 2>/dev/null
done
NS2_LIB=$L timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k attention > $O/t.txt 2>&1; tail -1 $O/t.txt
echo NOSLP; cat $O/att_noslp.txt; echo SLP; cat $O/att_slp.txt
