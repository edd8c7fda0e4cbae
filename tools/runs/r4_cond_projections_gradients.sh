#!/bin/bash
# batched conditioning projections of the training forward: model-level backward tests + the warm training step
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r4o; rm -rf $O; mkdir -p $O
timeout 100 python -m pytest tests/test_backward_gpu.py -q -m gpu --tb=short -x -k "every_gradient or conditioned or ragged or stochastic or reference_wrapper or optimizer or allreducer" > $O/t_bwd.txt 2>&1; echo "backward rc=$?" | tee $O/summary.txt; tail -4 $O/t_bwd.txt
timeout 40 python tools/bench_train.py --shapes d512 --backends hip --iters 4 2>/dev/null | tee $O/train.txt | cut -c1-300
