#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4g; mkdir -p $O
timeout 900 python -m pytest tests/test_round4_gpu.py -q -m gpu --tb=short -k "config1 or checksum or data_writes" > $O/t_round4.txt 2>&1; echo "round4 rc=$?" >> $O/summary.txt
timeout 900 python -m pytest tests/test_parity_r2_gpu.py -q -m gpu --tb=short -k "large_activation" -s > $O/t_stress.txt 2>&1; echo "stress rc=$?" >> $O/summary.txt
cat $O/summary.txt; tail -5 $O/t_round4.txt; tail -5 $O/t_stress.txt
