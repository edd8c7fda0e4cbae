#!/bin/bash
# MFMA utilisation, wave stall split, L2 hit rate and HBM traffic of every GEMM / attention kernel of the timed bench.py step
# (the only MFMA-busy pass was round 1's).  Separate --pmc passes (MI355X_MICROARCH.md: SQ 8 slots, TCC 4,
# FETCH_SIZE / WRITE_SIZE one pass each), --kernel-trace only.  Writes gpurun_out/pmc_mfma_<precision>.json (copy to profiles/).
# usage (GPU box): tools/pmc_mfma_bench.sh <hybrid|mixed|exact|half> [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
PREC=${1:-hybrid}; shift
OUT=$R/gpurun_out/pmc_mfma_$PREC
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-side --no-parity --precision $PREC > $OUT/$n.log 2>&1
}
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE
run wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run tcc TCC_HIT_sum TCC_MISS_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
python - <<PY
import csv, glob, json, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for kind in ("mfma", "wait", "tcc", "fetch", "write"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % kind):
        for r in csv.DictReader(open(f)):
            res[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in res.items():
    if not any(n in k for n in ("gemm2_kernel", "gemm3_kernel", "ffconv3_kernel", "wavenet3_kernel", "attn_kernel", "rmsnorm_kernel")): continue
    m = {c: sum(v) / len(v) for c, v in d.items()}
    e = dict(launches=len(d.get("SQ_WAVE_CYCLES", d.get("FETCH_SIZE", [0]))), counters_mean_per_launch=m)
    if m.get("GRBM_GUI_ACTIVE"):
        act = m["GRBM_GUI_ACTIVE"] / 8.0                      # summed over the 8 XCDs
        e["mfma_pipe_busy_frac"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / act, 4)     # busy cycles summed over 1024 SIMDs
        e["cu_busy_frac"] = round(m.get("SQ_BUSY_CU_CYCLES", 0.0) / 256.0 / act, 4) if m.get("SQ_BUSY_CU_CYCLES") else None
    if m.get("SQ_WAVE_CYCLES") and m.get("SQ_WAIT_ANY") is not None:
        e["wave_cycles_split"] = dict(wait_any=round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 4) if "SQ_WAIT_ANY" in m else None)
    if m.get("SQ_WAIT_ANY") is not None and m.get("SQ_ACTIVE_INST_ANY") is not None:
        tot = m["SQ_WAIT_ANY"] + m.get("SQ_WAIT_INST_ANY", 0.0) + m["SQ_ACTIVE_INST_ANY"]
        e["wave_cycles_split"] = {k2: round(m.get(k2, 0.0) / tot, 4) for k2 in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")}
    if m.get("TCC_HIT_sum") is not None and m.get("TCC_MISS_sum") is not None and m["TCC_HIT_sum"] + m["TCC_MISS_sum"] > 0:
        e["l2_hit_rate"] = round(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 4)
    if "FETCH_SIZE" in m or "WRITE_SIZE" in m:
        e["hbm_bytes_per_launch_as_reported"] = int((m.get("FETCH_SIZE", 0.0) + m.get("WRITE_SIZE", 0.0)) * 1024)
        e["hbm_bytes_per_launch_read_side_doubled"] = int((2 * m.get("FETCH_SIZE", 0.0) + m.get("WRITE_SIZE", 0.0)) * 1024)
    out[k] = e
j = dict(note="rocprofv3 --pmc passes over bench.py --precision $PREC --steps 3 --warmup 1 (tools/pmc_mfma_bench.sh): means per launch over the "
              "launches of each pass; kernel durations under --pmc are inflated, use the counters as ratios.  mfma_pipe_busy_frac = "
              "SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); FETCH_SIZE is reported at half the bytes of wide "
              "coalesced streams on gfx950 (MI355X_MICROARCH.md, HBM) -- both forms given; L2 misses served by the Infinity Cache count "
              "as fabric reads.", precision="$PREC", kernels=out)
json.dump(j, open("$R/gpurun_out/pmc_mfma_$PREC.json", "w"), indent=1)
for k, e in out.items():
    print(k[:70], e.get("launches"), "mfma_busy", e.get("mfma_pipe_busy_frac"), "l2_hit", e.get("l2_hit_rate"), "hbm MB", round(e.get("hbm_bytes_per_launch_read_side_doubled", 0) / 1e6, 1))
PY
rm -rf $OUT/*/
