#!/bin/bash
# Round-2 GPU pass A1 (quick): scaled-MFMA probe, smoke, kernel-level tests at all precisions, short bench of three modes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2a1
rm -rf $OUT; mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
( time python tools/probe/run_scale_probe.py ) > $OUT/scale_probe.log 2>&1
( time python __graft_entry__.py smoke ) > $OUT/smoke.log 2>&1
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | tail -60 ) > $OUT/pytest_kernels.log 2>&1
for P in mixed half exact; do
  ( time python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-side --precision $P ) > $OUT/bench_$P.json 2> $OUT/bench_$P.err
done
tail -12 $OUT/scale_probe.log; tail -4 $OUT/smoke.log; tail -25 $OUT/pytest_kernels.log; for P in mixed half exact; do head -c 1200 $OUT/bench_$P.json; echo; tail -2 $OUT/bench_$P.err; done
