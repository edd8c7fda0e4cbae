#!/bin/bash
# Round 5, GPU call 1: (1) kernel + model parity of the new GEMM paths (W-request skip of the last column tile, dense half operands
# of the hybrid Wavenet), (2) A/B of both on the headline step, same box, alternating: new default | NS2_WAVENET_DENSE=0 | the
# no-W-skip build (tools/build_ab_variant.sh nowskip ...), (3) rocprofv3 kernel stats of the new default.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5a; rm -rf $O; mkdir -p $O
B="--steps 20 --warmup 3 --no-side --no-secondary --no-cpu-baseline --no-parity"
( timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x --tb=short -k "causal_conv or linear_f32 or geglu or qkv or wavenet or conv_k9 or conv_elu or linearity" 2>&1 | tail -8 ) > $O/t_kernels.txt; echo "kernels rc=${PIPESTATUS[0]}" >> $O/summary.txt
( timeout 900 python -m pytest tests/test_round5_gpu.py -q -m gpu --tb=short -s -k "wavenet or frozen" 2>&1 | tail -30 ) > $O/t_round5.txt
( timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu --tb=short 2>&1 | tail -8 ) > $O/t_model.txt
for rep in 1 2; do
  timeout 300 python bench.py $B > $O/bench_new_$rep.json 2> $O/bench_new_$rep.err
  NS2_WAVENET_DENSE=0 timeout 300 python bench.py $B > $O/bench_gather_$rep.json 2> $O/bench_gather_$rep.err
  NS2_LIB=$PWD/tools/ab/libns2hip_nowskip.so timeout 300 python bench.py $B > $O/bench_nowskip_$rep.json 2> $O/bench_nowskip_$rep.err
done
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in new gather; do
  if [ $v = gather ]; then export NS2_WAVENET_DENSE=0; else unset NS2_WAVENET_DENSE; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$v -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-side --no-parity > $R/$O/prof_$v.log 2>&1
  cp $(ls $R/$O/prof_$v/*/*kernel_stats.csv | head -1) $R/$O/bench_hybrid_${v}_kernel_stats.csv; rm -rf $R/$O/prof_$v
done
unset NS2_WAVENET_DENSE
cd $R
cp gpurun_out/parity_r5.json $O/ 2>/dev/null
tail -3 $O/t_kernels.txt $O/t_round5.txt $O/t_model.txt
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
for v in new gather; do echo == $v; head -12 $O/bench_hybrid_${v}_kernel_stats.csv | cut -d, -f1-5 | cut -c1-160; done
