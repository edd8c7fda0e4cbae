#!/bin/bash
# Round-2 GPU pass A: scaled-MFMA probe, smoke, the whole -m gpu suite (no -x: collect every failure), the default bench line,
# and rocprofv3 kernel stats of the mixed / half modes.  Outputs under gpurun_out/r2a/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2a
rm -rf $OUT; mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
( time python tools/probe/run_scale_probe.py ) > $OUT/scale_probe.log 2>&1
( time python __graft_entry__.py smoke ) > $OUT/smoke.log 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --ignore=tests/test_parity_r2_gpu.py 2>&1 | tail -40 ) > $OUT/pytest_old.log 2>&1
( time timeout 2400 python -m pytest tests/test_parity_r2_gpu.py -m gpu -q --timeout 1200 -p no:cacheprovider -rA 2>&1 | tail -120 ) > $OUT/pytest_parity.log 2>&1
( time python bench.py --steps 20 --warmup 3 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
cd /tmp && export TMPDIR=/tmp
for P in mixed half; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$P -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-side --precision $P > $OUT/prof_$P.log 2>&1
  cp $(ls $OUT/prof_$P/*/*kernel_stats.csv | head -1) $OUT/bench_${P}_kernel_stats.csv
  rm -rf $OUT/prof_$P
done
cd $R
tail -3 $OUT/scale_probe.log; tail -3 $OUT/smoke.log; tail -5 $OUT/pytest_old.log; tail -15 $OUT/pytest_parity.log; head -c 1500 $OUT/bench_default.json
