#!/bin/bash
# effective clock of the FF-conv GEMM micro-benchmark: GRBM_GUI_ACTIVE (per-XCD cycles, summed over 8 XCDs) / kernel time
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for d in 0; do
  OUT=$R/gpurun_out/clk_$d; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT -- python $R/tools/bench_gemm.py --which ffconv --prec 3 --iters 5 > $OUT/log 2>&1
  python - <<PY
import csv, glob
cc = glob.glob("$OUT/*/*counter_collection.csv")[0]; kt = glob.glob("$OUT/*/*kernel_trace.csv")[0]
dur = {}
for r in csv.DictReader(open(kt)):
    if "gemm2" in r["Kernel_Name"]: dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
vals = {}
for r in csv.DictReader(open(cc)):
    if "gemm2" in r["Kernel_Name"]: vals.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0); vals[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
ids = sorted(dur)[-4:]
ns = sum(dur[i] for i in ids) / len(ids)
gui = sum(vals["GRBM_GUI_ACTIVE"][i] for i in ids) / len(ids)
mf = sum(vals["SQ_VALU_MFMA_BUSY_CYCLES"][i] for i in ids) / len(ids)
print(f"dbg=$d  kernel {ns/1e3:8.1f} us  GRBM_GUI_ACTIVE/8 = {gui/8:10.0f} cyc -> {gui/8/ns:5.2f} GHz   MFMA busy = {mf/1024:10.0f} cyc/SIMD = {100*mf/1024/(gui/8):5.1f} % of active")
PY
done
