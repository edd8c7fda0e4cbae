#!/bin/bash
# codec side: kernel tests, codec parity tests, timings (fused residual block A/B)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3k
rm -rf $OUT; mkdir -p $OUT
cd $R
( timeout 400 python -m pytest tests/test_parity_r2_gpu.py tests/test_kernels_gpu.py -q -x -k "lstm or elu or prep2" -p no:cacheprovider 2>&1 | tail -15 ) > $OUT/pytest_lstm.log 2>&1
cat $OUT/pytest_lstm.log
for fe in 1 0 1; do
  echo "== NS2_SEANET_FUSED_RESBLOCK=$fe"
  NS2_SEANET_FUSED_RESBLOCK=$fe timeout 200 python tools/run_codec.py --decode --iters 5 2>&1 | tail -1
done | tee $OUT/codec_ab.log
( timeout 400 python -m pytest tests -q -m gpu -k "seanet or encodec or codec" -p no:cacheprovider 2>&1 | tail -6 ) > $OUT/pytest_codec.log 2>&1
cat $OUT/pytest_codec.log
