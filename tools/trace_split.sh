#!/bin/bash
# rocprofv3 kernel trace of the f32_split train step with the weight-gradient stream OFF (every duration is the kernel's own),
# summarised per kernel and step:   tools/trace_split.sh tag [mode] [extra bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; MODE=${2:-f32_split}; shift; shift
export SAR_WGRAD_STREAM=${SAR_WGRAD_STREAM:-0}
O=$R/gpurun_out/trace_split_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --mfma $MODE --steps 5 --warmup 2 --no-cpu-baseline --no-isolated-pass --no-secondary --warm-seconds 0 --sustained-steps 0 "$@" > $O/bench.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not f:
    print(open(sys.argv[1] + "/bench.log").read()[-1500:]); sys.exit(0)
tot = 0
rows = []
for r in csv.DictReader(open(f[0])):
    n = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n)[:60]
    ms = float(r["TotalDurationNs"]) / 1e6 / 7
    rows.append((ms, n, int(r["Calls"]) / 7, float(r["AverageNs"]) / 1e3))
    tot += ms
for ms, n, c, avg in sorted(rows, reverse=True)[:40]:
    print("%-60s %5.1f/step avg %8.1f us %7.3f ms/step" % (n, c, avg, ms))
print("TOTAL kernel ms/step %.3f" % tot)
PY
tail -1 $O/bench.log | cut -c1-200
