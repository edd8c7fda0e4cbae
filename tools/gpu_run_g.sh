#!/bin/bash
# Round-2 GPU pass G: the hybrid plan (mixed everywhere, FF causal conv as one half product) and the bf16x3 cross attention:
# smoke, the model-level suites, the whole parity file.  Outputs under gpurun_out/r2g/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2g
rm -rf $OUT; mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
( time python __graft_entry__.py smoke ) > $OUT/smoke.log 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --ignore=tests/test_parity_r2_gpu.py --ignore=tests/test_kernels_gpu.py 2>&1 | tail -40 ) > $OUT/pytest_model.log 2>&1
( time timeout 2400 python -m pytest tests/test_parity_r2_gpu.py -m gpu -q --timeout 1200 -p no:cacheprovider -rA 2>&1 | tail -150 ) > $OUT/pytest_parity.log 2>&1
tail -3 $OUT/smoke.log; tail -5 $OUT/pytest_model.log; tail -25 $OUT/pytest_parity.log
