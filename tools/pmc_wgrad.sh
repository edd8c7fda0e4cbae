#!/bin/bash
# Counters of the two weight-gradient GEMMs (transposed-copy route: gemm2_kernel<.., false>; row-plane route: <.., true>) on one shape
# of tools/bench_wgrad.py: wave-cycle split, instruction counts, L2 hits, HBM bytes.   usage: tools/pmc_wgrad.sh <shape substring> [precision]
R=${GRAFT_REPO_ROOT:-$(pwd)}
SH=${1:-ff_conv}; PREC=${2:-4}
OUT=$R/gpurun_out/pmc_wgrad; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -- python $R/tools/bench_wgrad.py --precision $PREC --iters 3 --only $SH > $OUT/$n.log 2>&1; }
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE
run wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run inst SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run tcp TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_UTCL1_TRANSLATION_MISS_sum
run fetch FETCH_SIZE
python - <<PY
import csv, glob, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gemm2_kernel" in r["Kernel_Name"]:
            res["TR" if "true>(" in r["Kernel_Name"].replace(", true, 0, true>", ", X, 0, true>(") else "plain"][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted(set(res["TR"]) | set(res["plain"]))
print("%-34s %16s %16s %8s" % ("counter (mean per launch)", "transposed-copy", "row-plane (TR)", "ratio"))
for c in names:
    a = sum(res["plain"][c]) / max(1, len(res["plain"][c])); b = sum(res["TR"][c]) / max(1, len(res["TR"][c]))
    print("%-34s %16.4g %16.4g %8.3f" % (c, a, b, b / a if a else float("nan")))
PY
rm -rf $OUT/*/
