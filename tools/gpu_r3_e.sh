#!/bin/bash
# pass E: failing tests with full tracebacks, attention workgroup-size A/B, then pass D (phased conv3)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3e
rm -rf $OUT; mkdir -p $OUT
cd $R
( timeout 600 python -m pytest tests/test_model_gpu.py tests/test_reference_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "determinism or sharded_sampler or compat_route or attention" 2>&1 | tail -60 ) > $OUT/pytest_fail.log 2>&1
cat $OUT/pytest_fail.log | cut -c1-300
( timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "attention" 2>&1 | tail -5 ) 
for NW in 4 8; do NS2_ATTN_NW=$NW python tools/bench_attention.py 2>/dev/null | tail -1; done
bash tools/gpu_r3_d.sh
