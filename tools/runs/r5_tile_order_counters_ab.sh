#!/bin/bash
# Round 5, GPU call 3: (1) A/B on one box, alternating: new default (helper loop + column-grouped tile order) | NS2_COL_GROUP=0 | the
# no-skip build; (2) PMC passes (MFMA busy, stall split, L2 hit rate, HBM traffic) of the step with and without the column-grouped
# order; (3) rocprofv3 kernel stats of a warm training step, exact vs mixed arithmetic; (4) the fixed mixed grad_prep test.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5c; rm -rf $O; mkdir -p $O
B="--steps 20 --warmup 3 --no-side --no-secondary --no-cpu-baseline --no-parity"
( timeout 600 python -m pytest tests/test_round5_gpu.py -q -m gpu --tb=short -k "grad_prep or planes_transpose or mixed_wgrad" 2>&1 | tail -n 8 ) > $O/t_round5.txt
( timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x --tb=short -k "causal_conv or geglu or linear_f32" 2>&1 | tail -n 5 ) > $O/t_kernels.txt
for rep in 1 2 3; do
  timeout 300 python bench.py $B > $O/bench_new_$rep.json 2> $O/bench_new_$rep.err
  NS2_COL_GROUP=0 timeout 300 python bench.py $B > $O/bench_nogroup_$rep.json 2> $O/bench_nogroup_$rep.err
  NS2_LIB=$PWD/tools/ab/libns2hip_nowskip.so timeout 300 python bench.py $B > $O/bench_nowskip_$rep.json 2> $O/bench_nowskip_$rep.err
done
timeout 900 tools/pmc_mfma_bench.sh hybrid > $O/pmc_new.log 2>&1; cp gpurun_out/pmc_mfma_hybrid.json $O/pmc_mfma_hybrid_new.json
NS2_COL_GROUP=0 timeout 900 tools/pmc_mfma_bench.sh hybrid > $O/pmc_nogroup.log 2>&1; cp gpurun_out/pmc_mfma_hybrid.json $O/pmc_mfma_hybrid_nogroup.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
for tp in exact mixed; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_train_$tp -- python $R/tools/bench_train.py --shapes d512 --backends hip --train-precision $tp --iters 3 --fused-adam > $R/$O/prof_train_$tp.log 2>&1
  cp $(ls $R/$O/prof_train_$tp/*/*kernel_stats.csv | head -1) $R/$O/train_d512_${tp}_kernel_stats.csv; rm -rf $R/$O/prof_train_$tp
done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_new -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-side --no-parity > $R/$O/prof_new.log 2>&1
cp $(ls $R/$O/prof_new/*/*kernel_stats.csv | head -1) $R/$O/bench_hybrid_kernel_stats.csv; rm -rf $R/$O/prof_new
cd $R
for f in t_round5 t_kernels; do echo "== $f"; tail -n 4 $O/$f.txt | cut -c1-300; done
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
tail -n 12 $O/pmc_new.log | cut -c1-200; tail -n 12 $O/pmc_nogroup.log | cut -c1-200
grep -h ms_per_step $O/prof_train_*.log | cut -c1-260
