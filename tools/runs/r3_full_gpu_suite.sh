#!/bin/bash
# full GPU test suite + parity record + default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3full
rm -rf $OUT; mkdir -p $OUT
cd $R
rm -f gpurun_out/parity_r3.json
( timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 -p no:cacheprovider ${PYTEST_ARGS:-} 2>&1 | tail -40 ) > $OUT/pytest_gpu.log 2>&1
tail -25 $OUT/pytest_gpu.log
cp gpurun_out/parity_r3.json $OUT/parity_r3.json 2>/dev/null
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
