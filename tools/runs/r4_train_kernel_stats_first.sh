#!/bin/bash
# round 4, third GPU call: round-4 tests again, RCCL world-1 reducer, kernel profile of a warm training step
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4c; mkdir -p $O
timeout 900 python -m pytest tests/test_round4_gpu.py -q -m gpu --tb=short  > $O/t_round4.txt 2>&1; echo "round4 rc=$?" >> $O/summary.txt
timeout 600 python -m pytest tests/test_backward_gpu.py -q -m gpu --tb=short -k "rccl" > $O/t_rccl.txt 2>&1; echo "rccl rc=$?" >> $O/summary.txt
export TMPDIR=/tmp
for shp in d512 d128; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$shp -- python tools/bench_train.py --shapes $shp --backends hip --iters 3 > $O/prof_$shp.txt 2>&1
  f=$(find $O/prof_$shp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/train_${shp}_kernel_stats.csv
  rm -rf $O/prof_$shp
done
cat $O/summary.txt
head -30 $O/train_d512_kernel_stats.csv | cut -c1-220
