#!/bin/bash
# End-of-round measurement on the GPU box: bench lines, rocprofv3 kernel stats of the same commands, PMC HBM traffic.
# Outputs under gpurun_out/final/ (copy what should be judged into profiles/).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
cd $R
python bench.py --steps 20 --warmup 3 > $OUT/bench_exact.json 2> $OUT/bench_exact.err
python bench.py --steps 20 --warmup 3 --precision fast --no-cpu-baseline > $OUT/bench_fast.json 2> $OUT/bench_fast.err
python tools/bench_rvq.py > $OUT/bench_rvq.json 2> $OUT/bench_rvq.err
python tools/bench_power.py > $OUT/bench_power.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for P in exact fast; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$P -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --precision $P > $OUT/prof_$P.log 2>&1
  cp $(ls $OUT/prof_$P/*/*kernel_stats.csv | head -1) $OUT/bench_${P}_kernel_stats.csv
done
cd $R
tools/pmc_bench.sh > $OUT/pmc.log 2>&1
cp gpurun_out/pmc_traffic.json $OUT/pmc_traffic.json
rm -rf $OUT/prof_exact $OUT/prof_fast
head -c 600 $OUT/bench_exact.json; echo; head -c 300 $OUT/bench_fast.json; echo; cat $OUT/bench_rvq.json | head -c 400; echo; cat $OUT/bench_power.txt | grep prec=
