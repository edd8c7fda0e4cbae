#!/bin/bash
# Round 6: first product run of the lean mixed linear kernel (gemm3): tests, bench A/B (NS2_GEMM=4 = both round-6 kernels off)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r6d; rm -rf $O; mkdir -p $O
( timeout 900 python -m pytest tests/test_round6_gpu.py -q -m gpu --tb=short 2>&1 | tail -n 25 ) > $O/t_round6.txt
B="--steps 20 --warmup 3 --no-side --no-secondary --no-cpu-baseline --no-parity"
for rep in 1 2 3; do
  timeout 300 python bench.py $B > $O/bench_new_$rep.json 2> $O/bench_new_$rep.err
  NS2_GEMM=4 timeout 300 python bench.py $B > $O/bench_old_$rep.json 2> $O/bench_old_$rep.err
done
timeout 300 python bench.py $B --precision mixed > $O/bench_mixed_new.json 2> $O/bench_mixed_new.err
NS2_GEMM=4 timeout 300 python bench.py $B --precision mixed > $O/bench_mixed_old.json 2> $O/bench_mixed_old.err
for f in t_round6; do echo "== $f"; cat $O/$f.txt | cut -c1-250; done
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])
except Exception as e:
    print(sys.argv[1], "unreadable", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
