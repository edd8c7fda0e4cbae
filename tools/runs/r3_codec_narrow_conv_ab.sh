#!/bin/bash
# the 1-channel ends of the SEANet stacks on the vector ALUs: tests, codec parity, A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3m; rm -rf $OUT; mkdir -p $OUT
cd $R
( timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_parity_r2_gpu.py -q -k "narrow or seanet or encodec or codec" -p no:cacheprovider 2>&1 | tail -8 ) > $OUT/pytest.log 2>&1
cat $OUT/pytest.log
for v in 1 0 1; do
  echo "== NS2_SEANET_NARROW_CONV=$v"
  NS2_SEANET_NARROW_CONV=$v timeout 200 python tools/run_codec.py --decode --iters 5 2>&1 | tail -1
done | tee $OUT/codec_ab.log
