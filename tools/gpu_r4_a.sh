#!/bin/bash
# round 4, first GPU call: the backward kernels one group per process (a fault in one group must not hide the others), then a
# first warm train-step timing
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4a; mkdir -p $O
for k in grad_prep planes_transpose wgrad dgrad film_gate attention_forward weight_update; do
  timeout 600 python -m pytest tests/test_backward_gpu.py -q -m gpu -k "$k" --tb=short -x > $O/t_$k.txt 2>&1; echo "$k rc=$?" >> $O/summary.txt
done
for k in "d64_L2" "d128_L6" "d512_L12" conditioned reference_wrapper optimizer; do
  timeout 900 python -m pytest tests/test_backward_gpu.py -q -m gpu -k "$k" --tb=short -x -s > $O/t_$k.txt 2>&1; echo "$k rc=$?" >> $O/summary.txt
done
timeout 900 python tools/bench_train.py --shapes d128,d512_b8,d512 --iters 4 --out $O/train_step.json > $O/train_step.txt 2>&1
cat $O/summary.txt
tail -5 $O/train_step.txt
