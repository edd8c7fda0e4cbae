#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r4o
python tools/exp_wavenet_phase.py > gpurun_out/r4o/wavenet_phase.json 2> gpurun_out/r4o/err.txt; cat gpurun_out/r4o/wavenet_phase.json; tail -3 gpurun_out/r4o/err.txt
