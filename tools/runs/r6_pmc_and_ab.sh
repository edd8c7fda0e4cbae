#!/bin/bash
# Round 6: (1) same-box A/B of the hybrid step, alternating, the round-6 kernels on / off (NS2_GEMM=4); (2) PMC passes of the step
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r6f; rm -rf $O; mkdir -p $O
B="--steps 20 --warmup 3 --no-side --no-secondary --no-cpu-baseline --no-parity"
for rep in 1 2 3; do
  timeout 300 python bench.py $B > $O/bench_new_$rep.json 2> $O/bench_new_$rep.err
  NS2_GEMM=4 timeout 300 python bench.py $B > $O/bench_old_$rep.json 2> $O/bench_old_$rep.err
done
timeout 1200 tools/pmc_mfma_bench.sh hybrid > $O/pmc_new.log 2>&1; cp gpurun_out/pmc_mfma_hybrid.json $O/pmc_mfma_hybrid.json
timeout 300 python tools/bench_gemm3.py > $O/bench_gemm3.txt 2>&1
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
tail -n 14 $O/pmc_new.log | cut -c1-220; cat $O/bench_gemm3.txt
