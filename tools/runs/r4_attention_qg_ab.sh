#!/bin/bash
# attention variants A/B: NS2_ATTN_PK=1 = packed fp32 softmax arithmetic at 2 waves per SIMD
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4h2; mkdir -p $O
NS2_ATTN_PK=1 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -k "attention" > $O/t_attn.txt 2>&1; echo "attn(pk) rc=$?" >> $O/summary.txt
for rep in 1 2 3; do
  NS2_ATTN_PK=1 python tools/bench_attention.py >> $O/att_pk1.txt 2>/dev/null
  python tools/bench_attention.py >> $O/att_pk0.txt 2>/dev/null
done
cat $O/summary.txt; tail -2 $O/t_attn.txt; echo PK1; cat $O/att_pk1.txt; echo PK0; cat $O/att_pk0.txt
