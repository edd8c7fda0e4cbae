#!/bin/bash
# Round-2 GPU pass I: prepare_cond at bf16x3 for the IEEE-half modes -- error trace of the conditioned model, the model-level
# suites, the conditioned parity tests.  Outputs under gpurun_out/r2i/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2i
rm -rf $OUT; mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
( time python tools/cond_error_trace.py ) > $OUT/cond_trace.log 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --ignore=tests/test_parity_r2_gpu.py --ignore=tests/test_kernels_gpu.py 2>&1 | tail -30 ) > $OUT/pytest_model.log 2>&1
( time timeout 1500 python -m pytest tests/test_parity_r2_gpu.py -m gpu -q --timeout 900 -p no:cacheprovider -k "golden or conditioned or conditional or rvq or codec" 2>&1 | tail -30 ) > $OUT/pytest_parity.log 2>&1
cut -c1-420 $OUT/cond_trace.log; tail -4 $OUT/pytest_model.log; tail -6 $OUT/pytest_parity.log
