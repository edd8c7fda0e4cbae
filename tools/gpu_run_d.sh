#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2d
rm -rf $OUT; mkdir -p $OUT
cd $R
export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "rvq or skinny or time_embed or split or rmsnorm" 2>&1 | tail -15 ) > $OUT/pytest_k.log 2>&1
( timeout 1200 python -m pytest tests/test_parity_r2_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "rvq or encodec or stress or saturat or composition or ddim_traj" 2>&1 | tail -25 ) > $OUT/pytest_p.log 2>&1
python tools/bench_rvq.py > $OUT/rvq.txt 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-side > $OUT/bench_mixed.json 2> $OUT/bench_mixed.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-side > $OUT/prof.log 2>&1
cp $(ls $OUT/prof/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv; rm -rf $OUT/prof
cd $R
tail -6 $OUT/pytest_k.log; tail -12 $OUT/pytest_p.log; cat $OUT/rvq.txt | tail -5; head -c 600 $OUT/bench_mixed.json; echo; grep -i "skinny" $OUT/kernel_stats.csv | cut -c1-40,140-300
