#!/bin/bash
# Per-kernel PMC counters of the CN8 (bf16) conv kernels at the NTU layer shapes (GPU box).
#   tools/pmc8.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY ..." [kernels, e.g. t_fwd,t_dgrad] [lib.so]
# One rocprofv3 pass per call (--pmc with --kernel-trace only); prints the mean of every counter per kernel name and shape
# in launch order (kbench8.py walks the five layer shapes).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
[ -n "$3" ] && export SAR_HIP_LIB=$PWD/$3
OUT=gpurun_out/pmc8_$$
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc $1 -d $OUT -o p --output-format csv -- python3 tools/kbench8.py ${2:-t_fwd,t_dgrad} > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not f: print("no counter csv"); print(open(d + "/log.txt").read()[-2000:]); sys.exit(0)
kt = {r["Dispatch_Id"]: r for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]))}
agg = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    k = kt.get(r["Dispatch_Id"])
    if not k: continue
    dur = int(k["End_Timestamp"]) - int(k["Start_Timestamp"])
    if dur < 30000: continue
    name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0][:48]
    key = (name, r.get("Grid_Size", k.get("Grid_Size", "")))
    a = agg.setdefault(key, collections.defaultdict(list))
    a[r["Counter_Name"]].append(float(r["Counter_Value"]))
    a["_dur_us"].append(dur / 1e3)
for (name, grid), c in agg.items():
    print("%s grid=%s" % (name, grid))
    print("   " + "  ".join("%s=%.4g" % (cn, sum(v) / len(v)) for cn, v in sorted(c.items())))
PY
rm -rf $OUT
