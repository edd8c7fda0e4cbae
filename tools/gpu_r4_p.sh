#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r4p
timeout 900 python -m pytest tests/test_backward_gpu.py -q -m gpu --tb=short -k "odd_shapes" > gpurun_out/r4p/t.txt 2>&1; tail -40 gpurun_out/r4p/t.txt
