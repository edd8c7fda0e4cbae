#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4e; mkdir -p $O
timeout 900 python tools/measure_plan.py hybrid hybrid_ff mixed half --seeds 3 --out $O/plan_errors.json > $O/plan_errors.txt 2>&1
for rep in 1 2; do
  for P in hybrid hybrid_ff; do
    timeout 300 python bench.py --steps 20 --warmup 3 --no-side --no-secondary --no-cpu-baseline --no-parity --precision $P > $O/bench_${P}_$rep.json 2> $O/bench_${P}_$rep.err
  done
done
tail -30 $O/plan_errors.txt
for f in $O/bench_*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
