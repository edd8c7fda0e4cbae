#!/bin/bash
# Round 6: head dims 32 / 128 -- kernel test, model goldens (reference-generated) in every plan, the loud-error test
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r6h; rm -rf $O; mkdir -p $O
( timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_parity_r2_gpu.py -q -m gpu --tb=short -k "attention or golden or taps or loud" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -n 40 ) > $O/pytest_tail.txt
tail -n 25 $O/pytest_tail.txt | cut -c1-240
