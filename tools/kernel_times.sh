#!/bin/bash
# rocprofv3 kernel durations (no host gaps, no slab_reduce) of a kbench8 selection, per layer shape:  tools/kernel_times.sh g_wgrad [substring]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kb
rocprofv3 --kernel-trace --output-format csv -d /tmp/kb -- python3 $R/tools/kbench8.py $1 > /tmp/kb.log 2>&1
python3 - "$2" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/kb/**/*kernel_trace.csv", recursive=True)[0]
sub = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else "cn8"
rows = sorted((r for r in csv.DictReader(open(f)) if sub in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
runs, cur = [], []
for r in rows:          # kbench8 runs each case 13 times in a row
    cur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if len(cur) == 13:
        runs.append(cur); cur = []
for i, v in enumerate(runs):
    print("case %d: %s  min %.1f us  median %.1f us" % (i, rows[13 * i]["Kernel_Name"].replace("void (anonymous namespace)::", "")[:50], min(v), sorted(v)[6]))
PY
