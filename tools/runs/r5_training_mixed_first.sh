#!/bin/bash
# Round 5, GPU call 2: (1) the mixed training arithmetic: per-call kernels + every gradient vs the reference autograd; the resampler
# backward on the HIP Functions; the changed training ABI under the round-4 backward tests, (2) the helper loop of the tap-shared conv
# (parity + A/B against the no-skip build), (3) warm training-step timing: exact vs mixed, foreach vs fused Adam, HIP-graph replay.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5b; rm -rf $O; mkdir -p $O
B="--steps 20 --warmup 3 --no-side --no-secondary --no-cpu-baseline --no-parity"
( timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x --tb=short -k "causal_conv" 2>&1 | tail -n 5 ) > $O/t_conv.txt
( timeout 1200 python -m pytest tests/test_round5_gpu.py -q -m gpu --tb=short -s 2>&1 | tail -n 60 ) > $O/t_round5.txt
( timeout 1500 python -m pytest tests/test_backward_gpu.py -q -m gpu --tb=short 2>&1 | tail -n 40 ) > $O/t_backward.txt
for rep in 1 2; do
  timeout 300 python bench.py $B > $O/bench_new_$rep.json 2> $O/bench_new_$rep.err
  NS2_LIB=$PWD/tools/ab/libns2hip_nowskip.so timeout 300 python bench.py $B > $O/bench_nowskip_$rep.json 2> $O/bench_nowskip_$rep.err
done
timeout 900 python tools/bench_train.py --shapes d512 --backends hip --train-precision exact,mixed --iters 4 --out $O/train_d512.json > $O/train_d512.txt 2>&1
timeout 900 python tools/bench_train.py --shapes d512 --backends hip --train-precision exact,mixed --iters 4 --fused-adam --out $O/train_d512_fused.json > $O/train_d512_fused.txt 2>&1
timeout 600 python tools/bench_train.py --shapes d128 --backends hip --train-precision exact,mixed --iters 8 --fused-adam --out $O/train_d128.json > $O/train_d128.txt 2>&1
timeout 600 python tools/bench_train.py --shapes d128 --backends hip --train-precision exact,mixed --iters 8 --fused-adam --graph --out $O/train_d128_graph.json > $O/train_d128_graph.txt 2>&1
cp gpurun_out/parity_r5.json $O/ 2>/dev/null
for f in t_conv t_round5 t_backward; do echo "== $f"; tail -n 6 $O/$f.txt | cut -c1-300; done
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
grep -h "ms_per_step\|error" $O/train_*.txt | cut -c1-330
