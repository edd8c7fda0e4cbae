#!/bin/bash
# W padding rows from the zero page (one line request instead of one per row): A/B on one box, kernel tests with the new library
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4j; mkdir -p $O
ZP=$PWD/naturalspeech2_pytorch_amd/libns2hip_zp.so
NS2_LIB=$ZP timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short > $O/t_kernels.txt 2>&1; echo "kernels(zp) rc=$?" >> $O/summary.txt
NS2_LIB=$ZP timeout 900 python -m pytest tests/test_backward_gpu.py -q -m gpu --tb=short -k "wgrad or dgrad or d64_L2" > $O/t_backward.txt 2>&1; echo "backward(zp) rc=$?" >> $O/summary.txt
for rep in 1 2 3; do
  for P in hybrid exact; do
    NS2_LIB=$ZP timeout 300 python bench.py --steps 20 --warmup 3 --no-side --no-secondary --no-cpu-baseline --no-parity --precision $P > $O/bench_zp_${P}_$rep.json 2> $O/bench_zp_${P}_$rep.err
    timeout 300 python bench.py --steps 20 --warmup 3 --no-side --no-secondary --no-cpu-baseline --no-parity --precision $P > $O/bench_old_${P}_$rep.json 2> $O/bench_old_${P}_$rep.err
  done
done
cat $O/summary.txt; tail -2 $O/t_kernels.txt; tail -2 $O/t_backward.txt
for f in $O/bench_*.json; do echo -n "$f "; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"; done
