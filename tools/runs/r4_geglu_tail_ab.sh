#!/bin/bash
# GEGLU projection: partial last column tile on the 128x128 kernel (NS2_GEGLU_TAIL, default on) -- test + same-box A/B
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4i; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -k "geglu" > $O/t_geglu.txt 2>&1; echo "geglu rc=$?" >> $O/summary.txt
for rep in 1 2 3; do
  for v in 0 1; do
    for p in hybrid mixed exact; do
      echo -n "tail=$v $p " >> $O/ab.txt
      NS2_GEGLU_TAIL=$v python bench.py --steps 20 --warmup 5 --precision $p --no-side --no-secondary --no-cpu-baseline --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $O/ab.txt
    done
  done
done
cat $O/summary.txt; tail -3 $O/t_geglu.txt; cat $O/ab.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -k "wavenet" > $O/t_wn.txt 2>&1; echo "wavenet rc=$?"; tail -3 $O/t_wn.txt
timeout 900 python -m pytest tests -q -m gpu --tb=short -k "d128 or golden" > $O/t_d128.txt 2>&1; echo "d128 rc=$?"; tail -3 $O/t_d128.txt
for rep in 1 2; do python bench.py --steps 20 --warmup 5 --dim 128 --depth 6 --no-side --no-secondary --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | cut -c1-200; done
