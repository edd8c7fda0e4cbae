#!/bin/bash
# round 4, second GPU call: backward tests again (per-batch row planes fixed), round-4 inference tests, bench A/B of the hoisted
# time table, warm train-step timing
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4b; mkdir -p $O
for k in grad_prep attention_forward; do
  timeout 600 python -m pytest tests/test_backward_gpu.py -q -m gpu -k "$k" --tb=short > $O/t_$k.txt 2>&1; echo "$k rc=$?" >> $O/summary.txt
done
for k in "d64_L2" "d128_L6" "d512_L12" conditioned reference_wrapper optimizer; do
  timeout 900 python -m pytest tests/test_backward_gpu.py -q -m gpu -k "$k" --tb=short -s > $O/t_$k.txt 2>&1; echo "$k rc=$?" >> $O/summary.txt
done
timeout 900 python -m pytest tests/test_round4_gpu.py -q -m gpu --tb=short > $O/t_round4.txt 2>&1; echo "round4 rc=$?" >> $O/summary.txt
timeout 900 python tools/bench_train.py --shapes d128,d128_b32,d512_b8,d512 --backends hip --iters 4 --out $O/train_step.json > $O/train_step.txt 2>&1
for rep in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-side --no-secondary --no-cpu-baseline --no-parity > $O/bench_table_$rep.json 2> $O/bench_table_$rep.err
  timeout 300 python bench.py --steps 20 --warmup 3 --no-side --no-secondary --no-cpu-baseline --no-parity --no-time-table > $O/bench_notable_$rep.json 2> $O/bench_notable_$rep.err
done
cat $O/summary.txt
grep -h ms_per_step $O/train_step.txt | cut -c1-200
for f in $O/bench_*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
