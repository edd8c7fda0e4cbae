#!/bin/bash
# 96-byte fetch of the mixed-mode GEMMs: kernel tests, model parity, A/B against the 128-byte build on one box
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4f; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -x > $O/t_kernels.txt 2>&1; echo "kernels rc=$?" >> $O/summary.txt
timeout 1200 python -m pytest tests/test_model_gpu.py -q -m gpu --tb=short > $O/t_model.txt 2>&1; echo "model rc=$?" >> $O/summary.txt
timeout 600 python tools/measure_plan.py hybrid mixed --seeds 2 --out $O/plan_errors.json > $O/plan_errors.txt 2>&1
for rep in 1 2; do
  for P in hybrid mixed; do
    timeout 300 python bench.py --steps 20 --warmup 3 --no-side --no-secondary --no-cpu-baseline --no-parity --precision $P > $O/bench_new_${P}_$rep.json 2> $O/bench_new_${P}_$rep.err
    NS2_LIB=$PWD/naturalspeech2_pytorch_amd/libns2hip_h6off.so timeout 300 python bench.py --steps 20 --warmup 3 --no-side --no-secondary --no-cpu-baseline --no-parity --precision $P > $O/bench_old_${P}_$rep.json 2> $O/bench_old_${P}_$rep.err
  done
done
cat $O/summary.txt; tail -3 $O/t_kernels.txt; tail -3 $O/t_model.txt
python -c "
import json; d=json.load(open('$O/plan_errors.json'))
for k,v in d.items(): print(k, v)"
for f in $O/bench_*.json; do echo -n "$f "; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
