#!/bin/bash
# Round 6: the whole -m gpu suite on the final tree + smoke(), as the driver runs them.  -> profiles/r06_pytest_gpu_tail.txt, gpurun_out/parity_r6.json
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/full_r6a; rm -rf $O; mkdir -p $O
( timeout 2400 python -m pytest tests -q -m gpu --tb=short --durations=25 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -n 40 ) > $O/pytest_gpu_tail.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 3 ) > $O/smoke.txt
tail -n 12 $O/pytest_gpu_tail.txt | cut -c1-220; cat $O/smoke.txt
