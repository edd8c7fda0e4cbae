#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4n; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short 2>&1 | tail -6
timeout 900 python -m pytest tests/test_round4_gpu.py tests/test_model_gpu.py -q -m gpu --tb=short -k "split_k or golden or model" 2>&1 | tail -4
python tools/exp_small_m_kernel.py 2>/dev/null | tee $O/small_batch.json
