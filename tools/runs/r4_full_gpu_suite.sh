#!/bin/bash
# round 4: full GPU test suite + parity record + smoke (what the driver runs at round end)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4full
rm -rf $OUT; mkdir -p $OUT
cd $R
( timeout 2700 python -m pytest tests -m gpu -q --timeout 1500 -p no:cacheprovider ${PYTEST_ARGS:-} 2>&1 | tail -40 ) > $OUT/pytest_gpu.log 2>&1
tail -25 $OUT/pytest_gpu.log
cp gpurun_out/parity_r4.json $OUT/parity_r4.json 2>/dev/null
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
