#!/bin/bash
# Matrix-pipe utilisation of the split kernels from PMC counters (its own pass, no tracing): SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over
# the SIMDs: 32 per 32x32x16 MFMA) over GRBM_GUI_ACTIVE / 8 XCDs (the kernel's cycles) x 1 024 SIMDs.  Usage (GPU box): tools/mfma_busy.sh
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/mfma_busy
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT -o p --output-format csv -- python3 tools/kernel_bench.py --split f16x3a --reps 2 > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, re, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not f: print("no counter csv"); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "split" not in k and "ring" not in k: continue
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", k)
    name = m.group(1) if m else k[:50]
    key = (name, r["Grid_Size"])
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    agg[key]["n_" + r["Counter_Name"]] += 1
print("%-52s %-10s %6s %12s %10s" % ("kernel", "grid", "calls", "cycles", "MFMA busy"))
for (name, grid), v in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_VALU_MFMA_BUSY_CYCLES"]):
    n = max(v["n_GRBM_GUI_ACTIVE"], 1)
    cyc = v["GRBM_GUI_ACTIVE"] / 8 / n
    busy = v["SQ_VALU_MFMA_BUSY_CYCLES"] / max(v["n_SQ_VALU_MFMA_BUSY_CYCLES"], 1)
    print("%-52s %-10s %6d %12.0f %9.1f%%" % (name[:52], grid, n, cyc, 100.0 * busy / (cyc * 1024) if cyc else 0))
PY
