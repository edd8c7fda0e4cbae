#!/bin/bash
# A/B of the 128x128 GEMM's column sub-tile skip on the dim = 128 model (config 2): same box, alternating libraries
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3l; rm -rf $OUT; mkdir -p $OUT
cd $R
for lib in libns2hip.so libns2hip_noskip.so libns2hip.so libns2hip_noskip.so; do
  echo "== $lib"
  NS2_LIB=$R/naturalspeech2_pytorch_amd/$lib timeout 120 python bench.py --dim 128 --depth 6 --no-secondary --no-side --no-cpu-baseline --no-parity --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done | tee $OUT/ab.log
