#!/bin/bash
# HBM traffic of the dominant kernel during bench.py (separate --pmc passes, MI355X_MICROARCH.md §HBM): writes
# profiles-ready JSON to gpurun_out/pmc_traffic_<precision>.json.  usage (GPU box): tools/pmc_bench.sh <hybrid|mixed|exact|half|fast> [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
PREC=${1:-half}; shift
OUT=$R/gpurun_out/pmc_bench_$PREC
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-side --no-parity --precision $PREC "$@" > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-side --no-parity --precision $PREC "$@" > $OUT/write.log 2>&1
python - <<PY
import csv, glob, json, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for kind in ("fetch", "write"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % kind):
        for r in csv.DictReader(open(f)):
            res[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in res.items():
    if "gemm2_kernel" not in k and "attn_kernel" not in k: continue
    fs = d.get("FETCH_SIZE", [0]); ws = d.get("WRITE_SIZE", [0])
    out[k] = dict(launches=len(fs), fetch_kb_per_launch=sum(fs)/max(1,len(fs)), write_kb_per_launch=sum(ws)/max(1,len(ws)))
want = {"hybrid": "gemm2_kernel<1, 1, true,", "half": "gemm2_kernel<1, 1, true,", "mixed": "gemm2_kernel<2, 1, true,",
        "exact": "gemm2_kernel<3, 1, false,", "fast": "gemm2_kernel<1, 1, false,"}["$PREC"]
dom = [k for k in out if want in k]
j = dict(note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py; values are KB per launch as reported "
              "(gfx950: FETCH_SIZE under-reports wide coalesced streams by up to 2x, MI355X_MICROARCH.md HBM section; "
              "L2 misses served by the 256 MiB Infinity Cache are counted as fabric reads)", kernels=out)
if dom:
    k = dom[0]
    j["dominant_kernel"] = k
    j["hbm_bytes_per_launch_as_reported"] = int((out[k]["fetch_kb_per_launch"] + out[k]["write_kb_per_launch"]) * 1024)
    # gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of a wide coalesced stream (128-B requests
    # tallied at 64 B) -- the GEMM's operand reads are 16-B-per-lane LDS-DMA streams, so the read side is doubled
    j["hbm_bytes_per_launch"] = int((2 * out[k]["fetch_kb_per_launch"] + out[k]["write_kb_per_launch"]) * 1024)
j["precision"] = "$PREC"
json.dump(j, open("$R/gpurun_out/pmc_traffic_$PREC.json", "w"), indent=1)
print(json.dumps(j, indent=1)[:3000])
PY
