#!/bin/bash
# Round 6: upper bound of tap sharing in the Wavenet block's phase 1 (A rows requested once per 64-column chunk instead of once per tap):
# timing-only ablation build (results are garbage) against HEAD, per-kernel durations from rocprofv3 on the same box
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6wn; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in head ablate; do
  L=$R/naturalspeech2_pytorch_amd/libns2hip.so; [ $v = ablate ] && L=$R/naturalspeech2_pytorch_amd/libns2hip_ablate.so
  NS2_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -- python $R/bench.py --steps 6 --warmup 2 --no-side --no-secondary --no-cpu-baseline --no-parity > $O/$v.json 2> $O/$v.err < /dev/null
  f=$(ls $O/$v/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== $v"
  if [ -n "$f" ]; then cp $f $O/${v}_kernel_stats.csv; grep -E "wavenet3_kernel|ffconv3_kernel<|gemm3_kernel<3>" $f | cut -d, -f1-4 | cut -c1-120; fi
  rm -rf $O/$v
  python -c "import json;d=json.loads(open('$O/$v.json').read().strip().splitlines()[-1]);print('ms_per_step',d['ms_per_step'])" < /dev/null
done
