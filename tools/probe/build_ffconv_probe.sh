#!/bin/bash
# builds tools/probe/ffconv_probe (the stand-alone FF causal conv probe, round 6); temporaries (and the .s for reading) go to /tmp/ffc_tmp
set -e
cd "$(dirname "$0")"
mkdir -p /tmp/ffc_tmp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../naturalspeech2_pytorch_amd/csrc -I../../include -mllvm -pragma-unroll-threshold=200000 \
  ${FFC_DEFS} ffconv_probe.hip -o ${FFC_OUT:-ffconv_probe} -Rpass-analysis=kernel-resource-usage --save-temps=/tmp/ffc_tmp 2> /tmp/ffc_tmp/remarks.txt || { grep -E "error" -A3 /tmp/ffc_tmp/remarks.txt | head -40; exit 1; }
grep -E "Function Name|VGPRs:|SGPRs Spill|VGPRs Spill|ScratchSize" /tmp/ffc_tmp/remarks.txt | grep -A4 ffc_kernel | sed 's/\[-Rpass.*//; s/remark: ffconv_probe.hip:[0-9]*:0: *//' | paste - - - - - | sed 's/Function Name: //'
