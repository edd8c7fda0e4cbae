#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4n; rm -rf $O; mkdir -p $O
cd $R
python tools/exp_small_m_kernel.py 2>/dev/null | tee $O/small_batch.json
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_round4_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -4
