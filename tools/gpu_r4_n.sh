#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4n; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short 2>&1 | tail -3
timeout 900 python -m pytest tests -q -m gpu --tb=short -k "seanet or codec or split_k" 2>&1 | tail -3
python tools/exp_small_m_kernel.py 2>/dev/null | python -c "
import sys, json
d = json.load(sys.stdin)
print({k: v for k, v in d.items() if 'auto' in k})"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1 -- python $R/bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-side --no-parity > $O/prof_b1.log 2>&1
cp $(ls $O/prof_b1/*/*kernel_stats.csv | head -1) $O/bench_b1_hybrid_kernel_stats.csv; rm -rf $O/prof_b1
grep -E "finish|gemm_kernel" $O/bench_b1_hybrid_kernel_stats.csv | cut -c1-140
