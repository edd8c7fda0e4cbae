#!/bin/bash
# Round 6: the training pass as one HIP graph + the lean mixed linear kernel under the training packs (A/B), gradients vs the reference
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r6gt; rm -rf $O; mkdir -p $O
timeout 600 python tools/exp_graphed_train.py --headline --lean > $O/lean.txt 2>&1
timeout 600 python tools/exp_graphed_train.py --headline > $O/no_lean.txt 2>&1
( timeout 1200 python -m pytest tests/test_backward_gpu.py tests/test_round5_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -n 8 ) > $O/pytest_tail.txt
grep "^d\|^cond" $O/lean.txt | cut -c1-300; echo NO-LEAN; grep "^d\|^cond" $O/no_lean.txt | cut -c1-300; tail -4 $O/pytest_tail.txt
