#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2e
rm -rf $OUT; mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
( timeout 1200 python -m pytest tests/test_parity_r2_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "seanet or encodec or composition" 2>&1 | tail -60 ) > $OUT/pytest.log 2>&1
tail -60 $OUT/pytest.log
