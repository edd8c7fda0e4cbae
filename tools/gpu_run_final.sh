#!/bin/bash
# End-of-round pass: smoke, the whole -m gpu suite, then tools/final_profile_r2.sh (bench line, rocprofv3 stats per mode, PMC traffic).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2final
rm -rf $OUT; mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
( time python __graft_entry__.py smoke ) > $OUT/smoke.log 2>&1
( time timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 -p no:cacheprovider 2>&1 | tail -20 ) > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/smoke.log; tail -6 $OUT/pytest_gpu.log
bash tools/final_profile_r2.sh
