#!/bin/bash
# Per-kernel PMC counters of the conv kernels at chosen layers (GPU box).
#   tools/pmc_probe.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY ..." [lib.so]      env: KERNELS, LAYERS
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
[ -n "$2" ] && export SAR_HIP_LIB=$PWD/$2
OUT=gpurun_out/pmc_$$
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc $1 -d $OUT -o p --output-format csv -- python3 tools/kernel_bench.py --reps 2 --only ${KERNELS:-tconv_wgrad,gcn_wgrad} --layers ${LAYERS:-9} > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not f: print("no counter csv"); print(open(d + "/log.txt").read()[-2000:]); sys.exit(0)
kt = {r["Dispatch_Id"]: r for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]))}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = kt.get(r["Dispatch_Id"])
    if not k: continue
    dur = int(k["End_Timestamp"]) - int(k["Start_Timestamp"])
    if dur < 100000: continue
    name = r["Kernel_Name"].replace("void (anonymous namespace)::", "")[:70]
    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg[name]["_dur_us"].append(dur / 1e3)
for name, c in agg.items():
    print(name)
    for cn, v in sorted(c.items()):
        print("   %-28s %14.0f" % (cn, sum(v) / len(v)))
PY
rm -rf $OUT
