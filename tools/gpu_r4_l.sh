#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4l; mkdir -p $O
timeout 1200 python -m pytest tests/test_backward_gpu.py -q -m gpu --tb=short -k "every_gradient or conditioned or reference_wrapper or optimizer or stochastic or rccl" -s > $O/t_backward.txt 2>&1; echo "backward rc=$?" >> $O/summary.txt
timeout 900 python tools/bench_train.py --shapes d128,d512 --backends hip --iters 4 --out $O/train_step.json > $O/train_step.txt 2>&1
cat $O/summary.txt; grep -h ms_per_step $O/train_step.txt | cut -c1-220; tail -3 $O/t_backward.txt; grep "worst" $O/t_backward.txt | cut -c1-300
