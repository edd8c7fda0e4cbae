#!/bin/bash
# split-K of small products: kernel tests (third fixture variant), model-level test, small-batch timing
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r4m; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -k "splitk" > $O/t_kernels.txt 2>&1; echo "kernels[splitk] rc=$?" | tee -a $O/summary.txt; tail -15 $O/t_kernels.txt
timeout 1200 python -m pytest tests/test_round4_gpu.py -q -m gpu --tb=short -k "split_k" > $O/t_model.txt 2>&1; echo "model split_k rc=$?" | tee -a $O/summary.txt; tail -15 $O/t_model.txt
python tools/exp_small_m_kernel.py 2>/dev/null | tee $O/small_batch.json
