#!/bin/bash
# Effective GPU clock during each conv kernel: GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time (MI355X_MICROARCH.md,
# 'DVFS give-back').  Usage (GPU box): tools/clock_probe.sh [lib.so]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
[ -n "$1" ] && export SAR_HIP_LIB=$PWD/$1
OUT=gpurun_out/clock_$(basename ${1:-full} .so)
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $OUT -o p --output-format csv -- python3 tools/kernel_bench.py --reps 3 --only ${KERNELS:-tconv_fwd,tconv_dgrad} --layers ${LAYERS:-2,6,9} > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not f: print("no counter csv", glob.glob(d+"/**/*", recursive=True)); sys.exit(0)
rows = list(csv.DictReader(open(f[0])))
kt = {r["Dispatch_Id"]: r for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]))}
agg = collections.defaultdict(list)
for r in rows:
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    k = kt.get(r["Dispatch_Id"])
    if not k: continue
    dur = (int(k["End_Timestamp"]) - int(k["Start_Timestamp"])) * 1e-9
    if dur < 2e-4 or "conv_gemm" not in r["Kernel_Name"]: continue
    agg[(r["Kernel_Name"][:90], r["Grid_Size"])].append((float(r["Counter_Value"]) / 8 / dur / 1e9, dur * 1e3))
for k, v in agg.items():
    print("%-100s grid %-9s n=%d  clock %.2f GHz  %.3f ms" % (k[0], k[1], len(v), sorted(x[0] for x in v)[len(v)//2], sorted(x[1] for x in v)[len(v)//2]))
PY
