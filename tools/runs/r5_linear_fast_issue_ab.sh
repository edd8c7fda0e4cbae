#!/bin/bash
# Round 5: the "no zero page, no shift" DMA issue of plain linear products (gemm2.hip a_lin_full) against the general issue
# (build variant tools/ab/libns2hip_${VARIANT:-noalin}.so), alternating on one box; kernel tests on the new path first.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5j; rm -rf $O; mkdir -p $O
( timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x --tb=short 2>&1 | tail -n 4 ) > $O/t_kernels.txt
for rep in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-side --no-secondary --no-cpu-baseline --no-parity > $O/new_$rep.json 2> /dev/null
  NS2_LIB=$PWD/tools/ab/libns2hip_${VARIANT:-noalin}.so timeout 300 python bench.py --steps 20 --warmup 3 --no-side --no-secondary --no-cpu-baseline --no-parity > $O/old_$rep.json 2> /dev/null
done
for P in mixed; do
  timeout 300 python bench.py --steps 10 --warmup 3 --precision $P --no-side --no-secondary --no-cpu-baseline --no-parity > $O/new_$P.json 2> /dev/null
  NS2_LIB=$PWD/tools/ab/libns2hip_${VARIANT:-noalin}.so timeout 300 python bench.py --steps 10 --warmup 3 --precision $P --no-side --no-secondary --no-cpu-baseline --no-parity > $O/old_$P.json 2> /dev/null
done
tail -n 3 $O/t_kernels.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], j["ms_per_step"], j["roofline"]["achieved"], j["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
